"""The oracle's restatement of the frame pre-processing passes (BasicAPI.cpp:378-443, 506-636, 728-905) against
closed forms and an independent numpy evaluation.  The reference has no tests or vectors for these functions and
needs OpenCV / AVX2 hardware approximations the image cannot reproduce, so they stay "parity unpinned"
(oracle/tf_oracle.c says where); these tests pin the restatement to the formulas in the reference source."""
import numpy as np

from oracle import api as O
from texturefusion_amd import synth


def _cam(w=160, h=120):
    return synth.Camera(width=w, height=h, fx=131.25, fy=131.25, cx=79.5, cy=59.5)


def _plane_depth(cam, a, b, c):
    """z-depth image of the plane z = c + a X + b Y (camera frame): d = c / (1 - a x - b y), x = (j - cx) / fx ..."""
    j, i = np.meshgrid(np.arange(cam.width), np.arange(cam.height))
    x = (j - cam.cx) / cam.fx
    y = (i - cam.cy) / cam.fy
    return (c / (1.0 - a * x - b * y)).astype(np.float32)


def test_normal_map_of_planes_and_its_untouched_border():
    cam = _cam()
    for a, b in ((0.0, 0.0), (0.3, -0.2), (-0.5, 0.4)):
        d = _plane_depth(cam, a, b, 1.5)
        n = O.pre_normal_map(d, cam)
        want = np.array([-a, -b, 1.0]) / np.sqrt(a * a + b * b + 1.0)
        jlast = 1 + ((cam.width - 12) // 8) * 8
        inner = n[:, 1:cam.height - 1, 1:jlast + 8]
        # the cross product of the two central differences is the plane normal up to the discretisation of a
        # perspective image; the sign convention is (dX/dj) x (dX/di)
        err = np.abs(inner - want[:, None, None]).max()
        assert err < 2e-3, (a, b, err)
        assert (n[:, 0] == 0).all() and (n[:, -1] == 0).all() and (n[:, :, 0] == 0).all()
        assert (n[:, :, jlast + 8:] == 0).all(), "columns behind the last full 8-wide group are never written"


def test_normal_map_rejects_depth_steps_and_holes():
    cam = _cam()
    d = np.full((cam.height, cam.width), 1.0, np.float32)
    d[:, 80:] = 1.5      # a 0.5 m step: |depth_r - depth_l| >= 0.3 at columns 79 and 80
    d[40, 30] = 0.0      # a hole: its four neighbours see a 1 m difference
    n = O.pre_normal_map(d, cam)
    assert (n[:, 5:100, 79] == 0).all() and (n[:, 5:100, 80] == 0).all()
    assert (n[:, 40, 29] == 0).all() and (n[:, 40, 31] == 0).all() and (n[:, 39, 30] == 0).all()
    assert n[2, 60, 40] > 0.999


def test_refine_depth_by_normal_threshold():
    cam = _cam()
    d = np.full((cam.height, cam.width), 2.0, np.float32)
    n = np.zeros((3, cam.height, cam.width), np.float32)
    n[2] = 1.0                      # facing the camera: |view . n| ~ 1 -> kept
    n[:, 10, 10] = (1.0, 0.0, 0.0)  # grazing at the image centre column? no: view.x at j = 10 is -0.47 -> kept
    n[:, 60, 80] = (1.0, 0.0, 0.0)  # next to the principal point view = (0.004, 0.004, 1): |q| < 0.1 -> removed
    n2, d2 = O.pre_refine_depth_normal(n, d, cam)
    assert d2[60, 80] == 0 and (n2[:, 60, 80] == 0).all()
    assert d2[10, 10] == 2.0 and n2[0, 10, 10] == 1.0
    assert (d2 == 2.0).sum() == d.size - 1


def test_color_valid_and_quality_formulas():
    cam = _cam()
    rng = np.random.Generator(np.random.PCG64(5))
    n = rng.normal(size=(3, cam.height, cam.width)).astype(np.float32)
    n /= np.linalg.norm(n, axis=0, keepdims=True).astype(np.float32)
    j, i = np.meshgrid(np.arange(cam.width), np.arange(cam.height))
    v = np.stack([(j - cam.cx) / cam.fx, (i - cam.cy) / cam.fy, np.ones_like(j, np.float64)])
    v /= np.linalg.norm(v, axis=0, keepdims=True)
    q = np.abs((v * n).sum(0))
    flag = O.pre_color_valid(n, cam)
    sure = np.abs(q - 0.2) > 1e-5  # away from the threshold the f64 evaluation decides the same
    assert np.array_equal(flag[sure], (q >= 0.2)[sure].astype(np.uint8))
    assert set(np.unique(flag)) <= {0, 1}
    # quality = |Sobel_xy(gray)| * |view . n| where depth > 0, the raw derivative elsewhere
    rgb = rng.integers(0, 256, (cam.height, cam.width, 3), dtype=np.uint8)
    depth = np.where(rng.random((cam.height, cam.width)) < 0.8, 1.0, 0.0).astype(np.float32)
    got = O.pre_color_quality(depth, n, rgb, cam)
    r64 = rgb.astype(np.int64)
    gray = (4899 * r64[..., 0] + 9617 * r64[..., 1] + 1868 * r64[..., 2] + 8192) >> 14
    g = np.pad(gray, 1, mode="reflect")  # numpy "reflect" = BORDER_REFLECT_101
    s = (g[2:, 2:] - g[2:, :-2] - g[:-2, 2:] + g[:-2, :-2]).astype(np.float64)
    assert np.array_equal(got[depth == 0], s[depth == 0].astype(np.float32))
    want = np.abs(s) * q
    assert np.abs(got[depth > 0] - want[depth > 0]).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_gray_weights_are_opencv_8bit_fixed_point():
    cam = _cam()
    z = np.zeros((3, cam.height, cam.width), np.float32)
    d0 = np.zeros((cam.height, cam.width), np.float32)
    rgb = np.zeros((cam.height, cam.width, 3), np.uint8)
    rgb[50, 70] = (255, 255, 255)  # gray 255: (4899 + 9617 + 1868) * 255 + 8192 >> 14 = 255
    q = O.pre_color_quality(d0, z, rgb, cam)
    # the mixed derivative of a single bright pixel: +255 at (-1,-1) and (+1,+1), -255 at (-1,+1) and (+1,-1)
    assert q[49, 69] == 255 and q[51, 71] == 255 and q[49, 71] == -255 and q[51, 69] == -255
    assert np.count_nonzero(q) == 4
    rgb[50, 70] = (255, 0, 0)
    assert O.pre_color_quality(d0, z, rgb, cam)[49, 69] == 76  # (4899 * 255 + 8192) >> 14


def test_refine_newframe_identity_and_outliers():
    cam = _cam()
    d = _plane_depth(cam, 0.2, 0.1, 1.2)
    d[30:33, 40:44] = 0.0
    I = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    out = O.pre_refine_newframe(d, d, cam, I)
    inner = np.zeros_like(d, bool)
    inner[1:-1, 1:-1] = True
    # identity: every interior pixel finds itself (coordinate + 0.5 floors back), holes stay holes
    assert np.array_equal(out[inner & (d > 0)], d[inner & (d > 0)])
    assert (out[d == 0] == 0).all()
    far = d.copy()
    far[60:70, 60:70] *= 1.08  # 8 % off the keyframe's depth: rejected; 3 % stays
    far[80:90, 60:70] *= 1.03
    out = O.pre_refine_newframe(d, far, cam, I)
    assert (out[61:69, 61:69] == 0).all() and np.array_equal(out[81:89, 61:69], far[81:89, 61:69])


def test_refine_keyframe_running_mean_and_in_place_order():
    cam = _cam()
    d = _plane_depth(cam, 0.0, 0.0, 1.0)
    w = np.ones_like(d)
    I = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    new = (d * np.float32(1.02)).astype(np.float32)
    r, w2 = O.pre_refine_keyframe(d, w, new, cam, I)
    inner = np.zeros_like(d, bool)
    inner[3:-3, 3:-3] = True
    assert (w2[inner] == 2).all() and np.abs(r[inner] - 1.01).max() < 1e-6  # (1 * 1 + 1.02) / 2
    assert (w2[0] == 1).all() and np.array_equal(r[0], d[0])               # projections outside (2, W-2): untouched
    r1, w1, _ = in_place_chain_case(cam)
    # rows up to 52 see smooth taps (1.02): (1 + 1.02) / 2 = 1.01.  From row 54 on the taps fall into the checker,
    # the fallback reads the keyframe's own map three rows up -- a row the loop has ALREADY rewritten: 1.01, so
    # (1 + 1.01) / 2 = 1.005 (reading the original would give exactly 1), and three rows further down the
    # rewritten 1.005 is read: 1.0025, and so on down the image
    assert np.abs(r1[40, 20:140] - 1.01).max() < 1e-6
    assert np.abs(r1[54, 20:140] - 1.005).max() < 1e-6
    assert np.abs(r1[57, 20:140] - 1.0025).max() < 1e-6
    assert np.abs(r1[60, 20:140] - 1.00125).max() < 1e-6
    assert (w1[60, 20:140] == 2).all()


def in_place_chain_case(cam):
    """A keyframe / new-frame pair whose refinement has a dependency chain all the way down the image (also used by
    the device parity test: the device reaches the same result as a fixed point, one chain link per round)."""
    d = _plane_depth(cam, 0.0, 0.0, 1.0)
    w = np.ones_like(d)
    new = np.full_like(d, 1.02)
    chk = (1.0 + 0.2 * (np.indices(d.shape).sum(0) % 2)).astype(np.float32)
    new[50:] = chk[50:]
    T = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32)
    T[1, 3] = np.float32(-3.2 / cam.fy)  # at depth 1 the projection lands 3.2 rows up
    r, w2 = O.pre_refine_keyframe(d, w, new, cam, T)
    return r, w2, (d, w, new, T)


def test_loader_depth_pass_follows_the_published_bilateral_filter():
    """tfo_pre_frame_depth against the defining formula of the bilateral filter (Gaussian in space over the circular
    window, Gaussian in value, reflect-101 border) evaluated directly in f64: the restatement's 4096-bin table with
    linear interpolation stays within 2e-6 of it; the cut, the units and the u16 write-back are exact."""
    rng = np.random.default_rng(5)
    H, W = 48, 64
    z = (1500 + 300 * np.sin(np.arange(W) / 9.0)[None, :] + 400 * (np.arange(H)[:, None] > 24) + rng.normal(0, 4, (H, W)))
    z = z.astype(np.uint16)
    z[5:9, 5:9] = 0
    z[20, 30] = 9000
    for d in (9, 7):
        zo, ref = O.pre_frame_depth(z, 4.0, 1000.0, d)
        src = np.where(z.astype(np.float32) > np.float32(4.0) * np.float32(1000.0), 0, z).astype(np.float32) / np.float32(1000.0)
        r = d // 2
        pad = np.pad(src.astype(np.float64), r, mode="reflect")
        num, den = np.zeros((H, W)), np.zeros((H, W))
        for i in range(-r, r + 1):
            for j in range(-r, r + 1):
                if i * i + j * j > r * r:
                    continue
                v = pad[r + i:r + i + H, r + j:r + j + W]
                wgt = np.exp(-0.5 * (i * i + j * j) / 10.0 ** 2) * np.exp(-0.5 * (v - src) ** 2 / 0.03 ** 2)
                num += v * wgt
                den += wgt
        assert np.abs(ref - num / den).max() < 2e-6
        assert np.array_equal(zo, (ref * np.float32(1000.0)).astype(np.uint16))
        assert ref[6, 6] == 0.0 and zo[20, 30] < 4000  # a hole among holes stays a hole; the far reading was cut
    # constant and empty images are copied
    for const in (0, 777):
        zc = np.full((H, W), const, np.uint16)
        zo, ref = O.pre_frame_depth(zc, 4.0, 1000.0)
        assert np.array_equal(ref, np.full((H, W), np.float32(const) / np.float32(1000.0)))
