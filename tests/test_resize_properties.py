"""Properties of the atlas' resize branch (Atlas::UpdateBuffer, Structure/Atlas.cpp:66-96: a patch image larger than
its slot goes through cv::resize(..., INTER_LINEAR)).  OpenCV is not in the image, so the oracle restates the
published 8-bit algorithm (11-bit coefficients, two-pass fixed point); these are the size-independent facts that
restatement must satisfy whatever the source image:

  * a constant image resizes to the same constant (the coefficients of every tap pair sum to 2048),
  * an exact 2:1 reduction is the rounded mean of each 2x2 block, (a + b + c + d + 2) >> 2 -- OpenCV itself
    switches INTER_LINEAR to its INTER_AREA fast path at scale 2, and the two agree there,
  * an exact 1:1 "resize" never happens: a ROI that fits is copied (ratio stays 1),
  * the ratio written back is slot / ROI per axis; identical source rows come out within one count of each other.

CPU only; the GPU blit is compared with this oracle bit for bit in tests/test_gpu_atlas.py."""
import numpy as np
import pytest

from oracle import api as O

RES = np.float32(0.005)


def _atlas():
    a = O.Atlas(RES)
    assert a.pw == 24 and a.ph == 18  # 3 x 8 voxels ... the slot the tests below size their ROIs against
    return a


def _slot(a, texloc, w=None, h=None):
    w = a.pw if w is None else w
    h = a.ph if h is None else h
    x, y = texloc % a.w, texloc // a.w
    return a.buffer()[y:y + h, x:x + w]


@pytest.mark.parametrize("value", [(0, 0, 0), (255, 255, 255), (17, 200, 93)])
@pytest.mark.parametrize("roi", [(25, 19), (48, 36), (61, 18), (24, 55), (200, 137)])
def test_constant_image_stays_constant(value, roi):
    a = _atlas()
    img = np.empty((240, 320, 3), np.uint8)
    img[...] = np.asarray(value, np.uint8)
    ok, t = a.alloc()
    assert ok == 0
    cols, rows = roi
    r, ratio = a.blit(t, img, [7, 11, cols, rows])
    assert r == 0
    got = _slot(a, t)
    assert (got == np.asarray(value, np.uint8)).all()
    assert ratio[0] == (np.float32(a.pw) / np.float32(cols) if cols > a.pw else 1.0)
    assert ratio[1] == (np.float32(a.ph) / np.float32(rows) if rows > a.ph else 1.0)
    a.close()


def test_exact_two_to_one_is_the_rounded_block_mean():
    a = _atlas()
    rng = np.random.Generator(np.random.PCG64(11))
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    ok, t = a.alloc()
    bx, by = 13, 9
    r, ratio = a.blit(t, img, [bx, by, 2 * a.pw, 2 * a.ph])
    assert r == 0 and ratio[0] == 0.5 and ratio[1] == 0.5
    src = img[by:by + 2 * a.ph, bx:bx + 2 * a.pw].astype(np.int32)
    want = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    assert np.array_equal(_slot(a, t).astype(np.int32), want)
    a.close()


def test_two_to_one_along_one_axis_only():
    """cols = 2 PW, rows <= PH: cv::resize still fills the FULL slot (Atlas.cpp:86-90), so the row direction is an
    enlargement; along x every output pixel is the mean of its two source pixels of the interpolated row."""
    a = _atlas()
    rng = np.random.Generator(np.random.PCG64(12))
    img = rng.integers(0, 256, (100, 100, 3), dtype=np.uint8)
    for y in range(100):
        img[y] = img[0]  # rows identical: the vertical pass cannot change anything
    ok, t = a.alloc()
    r, ratio = a.blit(t, img, [3, 5, 2 * a.pw, a.ph])
    assert r == 0 and ratio[0] == 0.5 and ratio[1] == 1.0
    src = img[5, 3:3 + 2 * a.pw].astype(np.int32)
    want = (src[0::2] + src[1::2] + 1) >> 1
    got = _slot(a, t).astype(np.int32)
    assert np.array_equal(got, np.broadcast_to(want, got.shape))
    a.close()


def test_roi_that_fits_is_copied():
    a = _atlas()
    rng = np.random.Generator(np.random.PCG64(13))
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    ok, t = a.alloc()
    r, ratio = a.blit(t, img, [10, 20, a.pw, a.ph])
    assert r == 0 and ratio[0] == 1.0 and ratio[1] == 1.0
    assert np.array_equal(_slot(a, t), img[20:20 + a.ph, 10:10 + a.pw])
    ok, t2 = a.alloc()
    r, ratio = a.blit(t2, img, [1, 2, 5, 7])
    assert np.array_equal(_slot(a, t2, 5, 7), img[2:9, 1:6])
    a.close()


def test_column_constant_image_stays_column_constant():
    a = _atlas()
    img = np.zeros((90, 120, 3), np.uint8)
    img[...] = (np.arange(120) * 2 % 256).astype(np.uint8)[None, :, None]
    ok, t = a.alloc()
    r, _ = a.blit(t, img, [4, 4, 100, 70])
    assert r == 0
    got = _slot(a, t).astype(np.int32)
    # the vertical pass truncates its two products separately ((b0 * h) >> 16 + (b1 * h) >> 16), so identical
    # source rows may come out one count apart, never more
    assert np.abs(got - got[0:1]).max() <= 1
    # bilinear taps never overshoot: within a run without wrap-around the columns do not fall by more than that count
    d = np.diff(got[0, :, 0])
    assert (d[np.abs(d) < 100] >= -1).all()
    a.close()
