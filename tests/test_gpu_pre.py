"""Frame pre-processing on the device (tf_pre_*, texturefusion_amd/csrc/tf_pre.hip) against the oracle's restatement
of BasicAPI.cpp:378-905, bit for bit, on images resident in device memory: room frames with holes at 640x480 and
a small camera, a pair of frames related by a general rigid motion, and the in-place dependency chain of
refineKeyframesSIMD."""
import numpy as np
import pytest

from oracle import api as O
from texturefusion_amd import capi, synth
from tests.util import RES5, HipBuffer
from tests.test_oracle_pre import in_place_chain_case

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _dev(arr):
    arr = np.ascontiguousarray(arr)
    return HipBuffer(arr.nbytes).from_host(arr)


def _get(buf, dtype, shape):
    return buf.to_host().view(dtype).reshape(shape).copy()


def _rel(pose_a, pose_b):
    """f32 of (A^-1 * B)[3x4] for 3x4 poses A, B (the Sophus expression of BasicAPI.cpp:402-406 / :528-533, in f64)"""
    A = np.vstack([pose_a.astype(np.float64), [0, 0, 0, 1]])
    B = np.vstack([pose_b.astype(np.float64), [0, 0, 0, 1]])
    return (np.linalg.inv(A) @ B)[:3].astype(np.float32)


@pytest.mark.parametrize("size", [(640, 480), (160, 120)])
def test_frame_passes_match_the_oracle(gpu_required, size):
    w, h = size
    cam = synth.Camera() if w == 640 else synth.Camera(width=w, height=h, fx=131.25, fy=131.25, cx=79.5, cy=59.5)
    gv = capi.Volume(RES5, cam, max_chunks=1 << 10, max_list=1 << 10, max_coarse=1 << 12)
    d0, rgba0, _, pose0 = synth.room_frame(10, cam, with_quality=False, wobble=0.05)
    d1, rgba1, _, pose1 = synth.room_frame(12, cam, with_quality=False, wobble=0.05)
    shape = (cam.height, cam.width)
    # extractNormalMapSIMD
    n_o = O.pre_normal_map(d0, cam)
    bd, bn = _dev(d0), HipBuffer(n_o.nbytes)
    gv.pre_normal_map(bd.ptr, bn.ptr)
    gv.sync()
    n_g = _get(bn, np.float32, n_o.shape)
    assert np.array_equal(_bits(n_g), _bits(n_o))
    assert (np.linalg.norm(n_o, axis=0) > 0.9).mean() > 0.8
    # refineKeyframesSIMD (keyframe = frame 0 refined by frame 1), then refineNewframesSIMD the other way round
    wgt = np.ones(shape, np.float32)
    T_ref_new = _rel(pose1, pose0)
    r_o, w_o = O.pre_refine_keyframe(d0, wgt, d1, cam, T_ref_new)
    bw, bd1 = _dev(wgt), _dev(d1)
    rounds = gv.pre_refine_keyframe(bd.ptr, bw.ptr, bd1.ptr, T_ref_new)
    r_g, w_g = _get(bd, np.float32, shape), _get(bw, np.float32, shape)
    assert np.array_equal(_bits(r_g), _bits(r_o)) and np.array_equal(_bits(w_g), _bits(w_o)), rounds
    assert (w_o == 2).mean() > 0.5 and rounds >= 2
    T_new_ref = _rel(pose0, pose1)
    nw_o = O.pre_refine_newframe(r_o, d1, cam, T_new_ref)
    gv.pre_refine_newframe(bd.ptr, bd1.ptr, T_new_ref)
    gv.sync()
    assert np.array_equal(_bits(_get(bd1, np.float32, shape)), _bits(nw_o))
    assert 0.3 < (nw_o > 0).mean() < 1.0
    # refineDepthUseNormalSIMD on the keyframe (its normals were taken before the refinement, as in main.cpp)
    n2_o, d2_o = O.pre_refine_depth_normal(n_o, r_o, cam)
    gv.pre_refine_depth_normal(bn.ptr, bd.ptr)
    gv.sync()
    assert np.array_equal(_bits(_get(bn, np.float32, n_o.shape)), _bits(n2_o))
    assert np.array_equal(_bits(_get(bd, np.float32, shape)), _bits(d2_o))
    # checkColorQuality / estimateColorQuality
    rgb = np.ascontiguousarray(rgba0[..., :3])
    f_o = O.pre_color_valid(n2_o, cam)
    q_o = O.pre_color_quality(d2_o, n2_o, rgb, cam)
    bf, bq, brgb = HipBuffer(f_o.nbytes), HipBuffer(q_o.nbytes), _dev(rgb)
    gv.pre_color_valid(bn.ptr, bf.ptr)
    gv.pre_color_quality(bd.ptr, bn.ptr, brgb.ptr, bq.ptr)
    gv.sync()
    assert np.array_equal(_get(bf, np.uint8, shape), f_o)
    assert np.array_equal(_bits(_get(bq, np.float32, shape)), _bits(q_o))
    assert 0.5 < f_o.mean() < 1.0 and np.abs(q_o).max() > 10
    for b in (bd, bn, bw, bd1, bf, bq, brgb):
        b.free()
    gv.close()


def test_in_place_dependency_chain(gpu_required):
    """refineKeyframesSIMD rewrites the map its fallback reads: a chain of ~22 dependent rows down the image.  The
    device gets there as a fixed point and needs about one round per link."""
    cam = synth.Camera(width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5)
    r_o, w_o, (d, w, new, T) = in_place_chain_case(cam)
    gv = capi.Volume(RES5, cam, max_chunks=1 << 10, max_list=1 << 10, max_coarse=1 << 12)
    bd, bw, bn = _dev(d), _dev(w), _dev(new)
    rounds = gv.pre_refine_keyframe(bd.ptr, bw.ptr, bn.ptr, T)
    assert np.array_equal(_bits(_get(bd, np.float32, d.shape)), _bits(r_o))
    assert np.array_equal(_bits(_get(bw, np.float32, d.shape)), _bits(w_o))
    assert rounds >= 15
    for b in (bd, bw, bn):
        b.free()
    gv.close()


def _raw_depth(cam, k, depth_scale=1000.0):
    """a room frame as the sensor delivers it: u16 millimetres with holes, noise and a few readings beyond the cut"""
    d = synth.room_frame(k, cam, with_quality=False)[0]
    rng = np.random.default_rng(100 + k)
    z = d * depth_scale + rng.normal(0.0, 3.0, d.shape) * (d > 0)
    z[rng.random(d.shape) < 0.001] = 9000.0
    return np.clip(z, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize("size,d", [((640, 480), 9), ((640, 480), 7), ((160, 120), 9), ((40, 29), 5)])
def test_loader_depth_pass_matches_the_oracle(gpu_required, size, d):
    """DatasetWrapper::framePreprocess: maximum-depth cut, metres, cv::bilateralFilter, write-back -- bit for bit"""
    w, h = size
    cam = synth.Camera() if w == 640 else synth.Camera(width=w, height=h, fx=131.25 * w / 160, fy=131.25 * w / 160,
                                                        cx=w / 2 - 0.5, cy=h / 2 - 0.5)
    gv = capi.Volume(RES5, cam, max_chunks=1 << 10, max_list=1 << 10, max_coarse=1 << 12)
    z = _raw_depth(cam, 3)
    z_o, r_o = O.pre_frame_depth(z, 4.0, 1000.0, d)
    bz, br = _dev(z), HipBuffer(r_o.nbytes)
    gv.pre_frame_depth(bz.ptr, br.ptr, 4.0, 1000.0, d)
    gv.sync()
    assert np.array_equal(_bits(_get(br, np.float32, r_o.shape)), _bits(r_o))
    assert np.array_equal(_get(bz, np.uint16, z.shape), z_o)
    assert (z_o != z).mean() > 0.5 and z_o.max() <= 4000
    # a constant image is copied (max - min < FLT_EPSILON), and so is an empty one
    for const in (1234, 0):
        zc = np.full(z.shape, const, np.uint16)
        zc_o, rc_o = O.pre_frame_depth(zc, 4.0, 1000.0, d)
        bz.from_host(zc)
        gv.pre_frame_depth(bz.ptr, br.ptr, 4.0, 1000.0, d)
        gv.sync()
        assert np.array_equal(_bits(_get(br, np.float32, r_o.shape)), _bits(rc_o))
        assert np.array_equal(_get(bz, np.uint16, z.shape), zc_o)
    for b in (bz, br):
        b.free()
    gv.close()
