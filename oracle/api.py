"""ctypes binding of the CPU oracle (oracle/libtf_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from texturefusion_amd/.  See oracle/tf_oracle.h for the parity
status of the oracle itself.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TF_ORACLE_LIB points the tests at another build of the same sources (the sanitizer run: tools/oracle_asan.sh)
_LIB_PATH = os.environ.get("TF_ORACLE_LIB") or os.path.join(_HERE, "libtf_oracle.so")


def build(force: bool = False) -> str:
    if os.environ.get("TF_ORACLE_LIB"):
        return _LIB_PATH
    src = [os.path.join(_HERE, f) for f in ("tf_oracle.c", "tf_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Camera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("near_plane", C.c_float),
                ("far_plane", C.c_float)]


class Integrator(C.Structure):
    _fields_ = [("quad", C.c_float), ("lin", C.c_float), ("cons", C.c_float),
                ("scale", C.c_float), ("weight", C.c_float)]


class Keyframe(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("depth", C.POINTER(C.c_float)), ("T", C.c_float * 16),
                ("kf_id", C.c_int)]


class RowStats(C.Structure):
    _fields_ = [("rows_tsdf", C.c_int64), ("rows_color", C.c_int64),
                ("chunks_updated", C.c_int64)]


def default_integrator() -> Integrator:
    # MobileFusion::initChiselMap, GCFusion/MobileFusion.h:215-228
    return Integrator(np.float32(0.0019), np.float32(0.00152), np.float32(0.001504),
                      np.float32(6.0), np.float32(1.0))


def camera_from(cam) -> Camera:
    return Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.near, cam.far)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    u8p = C.POINTER(C.c_uint8)
    u16p = C.POINTER(C.c_uint16)
    i32p = C.POINTER(C.c_int32)
    i64p = C.POINTER(C.c_int64)
    u64p = C.POINTER(C.c_uint64)
    vp = C.c_void_p
    L.tfo_truncation.restype = C.c_float
    L.tfo_truncation.argtypes = [C.POINTER(Integrator), C.c_float]
    L.tfo_centroids.argtypes = [fp, C.c_float, fp]
    L.tfo_chunk_scalars.argtypes = [C.POINTER(Integrator), fp, i32p, C.c_float, fp, fp, fp]
    L.tfo_voxel_update.restype = C.c_int
    L.tfo_voxel_update.argtypes = [fp, u8p, fp, C.POINTER(Camera), C.POINTER(Integrator), fp,
                                   C.c_int, i32p, C.c_float, fp, fp, fp, u16p, fp,
                                   C.POINTER(RowStats)]
    L.tfo_bbox.argtypes = [fp, C.POINTER(Camera), fp, C.c_float, i32p, i32p]
    L.tfo_select.restype = C.c_int64
    L.tfo_select.argtypes = [fp, C.POINTER(Camera), C.POINTER(Integrator), fp, C.c_float, i32p,
                             C.c_int64, i64p]
    L.tfo_volume_create.restype = vp
    L.tfo_volume_create.argtypes = [C.c_float, C.c_int]
    L.tfo_volume_destroy.argtypes = [vp]
    L.tfo_volume_reset.argtypes = [vp]
    L.tfo_volume_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.tfo_volume_set_integrator.argtypes = [vp, C.POINTER(Integrator)]
    L.tfo_volume_set_threads.argtypes = [vp, C.c_int]
    L.tfo_volume_set_kernel.argtypes = [vp, C.c_int]
    L.tfo_have_avx2.restype = C.c_int
    L.tfo_set_sum_order.argtypes = [C.c_int]
    f4 = [C.c_float] * 4
    L.tfo_pre_normal_map.argtypes = [fp, C.c_int, C.c_int] + f4 + [fp]
    L.tfo_pre_refine_depth_normal.argtypes = [fp, fp, C.c_int, C.c_int] + f4
    L.tfo_pre_color_valid.argtypes = [fp, C.c_int, C.c_int] + f4 + [u8p]
    L.tfo_pre_color_quality.argtypes = [fp, fp, u8p, C.c_int, C.c_int] + f4 + [fp]
    L.tfo_pre_refine_newframe.argtypes = [fp, fp, C.c_int, C.c_int] + f4 + [fp]
    L.tfo_pre_refine_keyframe.argtypes = [fp, fp, fp, C.c_int, C.c_int] + f4 + [fp]
    L.tfo_pre_frame_depth.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_double,
                                      C.c_double, fp]
    L.tfo_get_sum_order.restype = C.c_int
    L.tfo_volume_num_chunks.restype = C.c_int64
    L.tfo_volume_num_chunks.argtypes = [vp]
    L.tfo_volume_list_chunks.restype = C.c_int64
    L.tfo_volume_list_chunks.argtypes = [vp, i32p, C.c_int64]
    L.tfo_volume_has_chunk.argtypes = [vp, i32p]
    L.tfo_volume_get_chunk.argtypes = [vp, i32p, fp, fp, u16p]
    L.tfo_volume_set_chunk.argtypes = [vp, i32p, fp, fp, u16p]
    L.tfo_set_select_kernel.argtypes = [C.c_int]
    L.tfo_set_select_kernel.restype = None
    L.tfo_volume_retract_observations.restype = C.c_int64
    L.tfo_volume_retract_observations.argtypes = [vp, C.c_int, i32p, C.c_int64]
    L.tfo_volume_get_observations.restype = C.c_int64
    L.tfo_volume_get_observations.argtypes = [vp, i32p, i32p, fp, C.c_int64]
    L.tfo_volume_num_dirty.restype = C.c_int64
    L.tfo_volume_num_dirty.argtypes = [vp]
    L.tfo_volume_list_dirty.restype = C.c_int64
    L.tfo_volume_list_dirty.argtypes = [vp, i32p, C.c_int64]
    L.tfo_volume_clear_dirty.argtypes = [vp]
    L.tfo_volume_get_rowstats.argtypes = [vp, C.POINTER(RowStats)]
    L.tfo_volume_clear_rowstats.argtypes = [vp]
    L.tfo_prepare.restype = C.c_int64
    L.tfo_prepare.argtypes = [vp, fp, fp, i32p, u8p, C.c_int64]
    L.tfo_integrate.restype = C.c_int
    L.tfo_integrate.argtypes = [vp, fp, u8p, fp, fp, i32p, C.c_int64, C.c_int, C.c_int, u8p, fp]
    L.tfo_finalize.restype = C.c_int64
    L.tfo_finalize.argtypes = [vp, i32p, u8p, u8p, C.c_int64, i32p]
    L.tfo_integrate_frame.restype = C.c_int64
    L.tfo_integrate_frame.argtypes = [vp, fp, u8p, fp, i64p]
    L.tfo_atlas_create.restype = vp
    L.tfo_atlas_create.argtypes = [C.c_float, C.c_int, C.c_int]
    L.tfo_atlas_destroy.argtypes = [vp]
    L.tfo_atlas_patch_w.argtypes = [vp]
    L.tfo_atlas_patch_h.argtypes = [vp]
    L.tfo_atlas_alloc.argtypes = [vp, u64p]
    L.tfo_atlas_loc_next.restype = C.c_uint64
    L.tfo_atlas_loc_next.argtypes = [vp]
    L.tfo_atlas_buffer.restype = C.POINTER(C.c_uint8)
    L.tfo_atlas_buffer.argtypes = [vp]
    L.tfo_patch_project.argtypes = [fp, fp, C.c_int64, fp, u8p, fp, C.POINTER(Camera), fp, fp,
                                    i32p, C.POINTER(C.c_int), i64p]
    L.tfo_atlas_blit.argtypes = [vp, C.c_uint64, u8p, C.c_int, C.c_int, i32p, fp]
    L.tfo_pack_vertices.argtypes = [C.c_int64, u8p, u8p, u8p, C.POINTER(C.c_uint64), fp, C.c_int, C.c_int, i64p, fp, fp,
                                    fp, fp, fp, fp, i64p, C.POINTER(C.c_uint32), fp, C.POINTER(C.c_uint32), i64p]
    L.tfo_pack_vertices.restype = C.c_int64
    L.tfo_patches_batch.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint64), i64p, fp, fp, fp, u8p, fp, C.POINTER(Camera), fp, fp]
    L.tfo_color_transfer.argtypes = [fp, fp, fp]
    L.tfo_color_compensate.argtypes = [C.c_int64, i32p, u8p, u8p, i64p, fp, fp, fp, fp, i32p]
    L.tfo_color_compensate.restype = C.c_int64
    L.tfo_atlas_hot_range.argtypes = [vp, u64p, C.c_int64, u64p, u64p]
    u32p = C.POINTER(C.c_uint32)
    L.tfo_mesh_chunk.restype = C.c_int64
    L.tfo_mesh_chunk.argtypes = [vp, i32p, fp, fp, fp, u32p, i64p]
    L.tfo_update_meshes.restype = C.c_int64
    L.tfo_update_meshes.argtypes = [vp]
    L.tfo_volume_num_meshes.restype = C.c_int64
    L.tfo_volume_num_meshes.argtypes = [vp]
    L.tfo_volume_list_meshes.restype = C.c_int64
    L.tfo_volume_list_meshes.argtypes = [vp, i32p, C.c_int64]
    L.tfo_volume_get_mesh.argtypes = [vp, i32p, i64p, i64p, fp, fp, fp, u32p, u8p, C.POINTER(C.c_int)]
    L.tfo_mesh_adjacency.argtypes = [fp, C.c_int64, fp, C.c_float, u8p]
    L.tfo_compress_meshes.restype = C.c_int64
    L.tfo_compress_meshes.argtypes = [vp, i32p, C.c_int64]
    L.tfo_generate_patches.argtypes = [vp, vp, i32p, C.c_int64, i32p, C.POINTER(Keyframe), u64p]
    L.tfo_update_atlas.argtypes = [vp, vp, i32p, C.c_int64]
    L.tfo_compensate_color_volume.restype = C.c_int64
    L.tfo_compensate_color_volume.argtypes = [vp]
    L.tfo_draw_meshes.restype = C.c_int64
    L.tfo_draw_meshes.argtypes = [vp, vp, fp, u32p, C.c_int64, C.c_int64, i64p]
    L.tfo_volume_get_patch.argtypes = [vp, i32p, u64p, C.POINTER(C.c_int), i32p, C.POINTER(C.c_int), fp, i64p, fp, fp, fp]
    L.tfo_frame_textured.restype = C.c_int64
    L.tfo_frame_textured.argtypes = [vp, vp, fp, u8p, fp, fp, C.c_int, u8p]
    _lib = L
    return L


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def truncation(ig: Integrator, z: float) -> float:
    return float(lib().tfo_truncation(C.byref(ig), np.float32(z)))


def weight(ig: Integrator, trunc: float) -> float:
    L = lib()
    L.tfo_weight.restype = C.c_float
    L.tfo_weight.argtypes = [C.POINTER(Integrator), C.c_float]
    return float(L.tfo_weight(C.byref(ig), np.float32(trunc)))


def chunk_scalars(ig: Integrator, pose, cid, res):
    """(originInCamera[3], truncation, weight) of one chunk (ProjectionIntegrator.cpp:74-101)."""
    pose = f32(pose).reshape(12)
    cid = np.ascontiguousarray(cid, np.int32)
    oc = np.zeros(3, np.float32)
    tr = C.c_float(0)
    w = C.c_float(0)
    lib().tfo_chunk_scalars(C.byref(ig), _p(pose, C.c_float), _p(cid, C.c_int32), np.float32(res),
                            _p(oc, C.c_float), C.byref(tr), C.byref(w))
    return oc, tr.value, w.value


def centroids(pose, res) -> np.ndarray:
    pose = f32(pose).reshape(12)
    out = np.empty(3 * 512, np.float32)
    lib().tfo_centroids(_p(pose, C.c_float), np.float32(res), _p(out, C.c_float))
    return out.reshape(3, 512)


def fresh_chunk():
    return (np.full(512, 999.0, np.float32), np.zeros(512, np.float32), np.zeros(2048, np.uint16))


def voxel_update(depth, rgba, quality, cam: Camera, ig: Integrator, pose, flag, cid, res,
                 sdf, weight, color, cen=None):
    """In-place update of (sdf, weight, color).  Returns (updated, quality, RowStats)."""
    pose = f32(pose).reshape(12)
    depth = f32(depth)
    if cen is None:
        cen = centroids(pose, res)
    cen = f32(cen).reshape(-1)
    cid = np.ascontiguousarray(cid, np.int32)
    q = C.c_float(0)
    st = RowStats()
    upd = lib().tfo_voxel_update(_p(depth, C.c_float), _p(rgba, C.c_uint8), _p(quality, C.c_float),
                                 C.byref(cam), C.byref(ig), _p(pose, C.c_float), int(flag),
                                 _p(cid, C.c_int32), np.float32(res), _p(cen, C.c_float),
                                 _p(sdf, C.c_float), _p(weight, C.c_float), _p(color, C.c_uint16),
                                 C.byref(q), C.byref(st))
    return bool(upd), q.value, st


def bbox(depth, cam: Camera, pose, res):
    pose = f32(pose).reshape(12)
    depth = f32(depth)
    mn = np.zeros(3, np.int32)
    mx = np.zeros(3, np.int32)
    lib().tfo_bbox(_p(depth, C.c_float), C.byref(cam), _p(pose, C.c_float), np.float32(res),
                   _p(mn, C.c_int32), _p(mx, C.c_int32))
    return mn, mx


def select(depth, cam: Camera, ig: Integrator, pose, res, cap=1 << 18):
    pose = f32(pose).reshape(12)
    depth = f32(depth)
    ids = np.zeros((cap, 3), np.int32)
    nc = C.c_int64(0)
    n = lib().tfo_select(_p(depth, C.c_float), C.byref(cam), C.byref(ig), _p(pose, C.c_float),
                         np.float32(res), _p(ids, C.c_int32), cap, C.byref(nc))
    if n > cap:
        return select(depth, cam, ig, pose, res, cap=int(n))
    return ids[:n].copy(), nc.value


class Volume:
    """Chisel + ChunkManager state of the oracle."""

    def __init__(self, res, cam=None, ig=None, use_color=True):
        self.L = lib()
        self.res = np.float32(res)
        self.h = self.L.tfo_volume_create(self.res, int(use_color))
        if cam is not None:
            self.set_camera(cam)
        if ig is not None:
            self.set_integrator(ig)

    def close(self):
        if self.h:
            self.L.tfo_volume_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.L.tfo_volume_reset(self.h)

    def set_camera(self, cam):
        self.cam = cam if isinstance(cam, Camera) else camera_from(cam)
        self.L.tfo_volume_set_camera(self.h, C.byref(self.cam))

    def set_integrator(self, ig):
        self.L.tfo_volume_set_integrator(self.h, C.byref(ig))

    def set_threads(self, n):
        self.L.tfo_volume_set_threads(self.h, int(n))

    def set_kernel(self, kernel):
        """0 = scalar restatement (checker), 1 = AVX2 row kernel (CPU baseline; bit-identical)."""
        self.L.tfo_volume_set_kernel(self.h, int(kernel))

    def num_chunks(self):
        return int(self.L.tfo_volume_num_chunks(self.h))

    def list_chunks(self):
        n = self.num_chunks()
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.L.tfo_volume_list_chunks(self.h, _p(ids, C.c_int32), n)
        return ids[:n]

    def has_chunk(self, cid):
        cid = np.ascontiguousarray(cid, np.int32)
        return bool(self.L.tfo_volume_has_chunk(self.h, _p(cid, C.c_int32)))

    def get_chunk(self, cid):
        cid = np.ascontiguousarray(cid, np.int32)
        sdf, w, col = fresh_chunk()
        r = self.L.tfo_volume_get_chunk(self.h, _p(cid, C.c_int32), _p(sdf, C.c_float),
                                        _p(w, C.c_float), _p(col, C.c_uint16))
        if r != 0:
            raise KeyError(tuple(cid))
        return sdf, w, col

    def set_chunk(self, cid, sdf, w, col):
        cid = np.ascontiguousarray(cid, np.int32)
        self.L.tfo_volume_set_chunk(self.h, _p(cid, C.c_int32), _p(f32(sdf), C.c_float),
                                    _p(f32(w), C.c_float),
                                    _p(np.ascontiguousarray(col, np.uint16), C.c_uint16))

    def observations(self, cid):
        cid = np.ascontiguousarray(cid, np.int32)
        kf = np.zeros(256, np.int32)
        q = np.zeros(256, np.float32)
        n = self.L.tfo_volume_get_observations(self.h, _p(cid, C.c_int32), _p(kf, C.c_int32),
                                               _p(q, C.c_float), 256)
        if n < 0:
            raise KeyError(tuple(cid))
        return {int(kf[i]): float(q[i]) for i in range(n)}

    def retract_observations(self, kf, ids):
        """MobileFusion::RetractObservations' chunk side: observations.erase(kf) for the listed chunks"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        return int(self.L.tfo_volume_retract_observations(self.h, int(kf), _p(ids, C.c_int32), len(ids)))

    def dirty(self):
        n = int(self.L.tfo_volume_num_dirty(self.h))
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.L.tfo_volume_list_dirty(self.h, _p(ids, C.c_int32), n)
        return ids[:n]

    def clear_dirty(self):
        self.L.tfo_volume_clear_dirty(self.h)

    def rowstats(self, clear=False):
        st = RowStats()
        self.L.tfo_volume_get_rowstats(self.h, C.byref(st))
        if clear:
            self.L.tfo_volume_clear_rowstats(self.h)
        return st

    def prepare(self, depth, pose, cap=1 << 18):
        depth = f32(depth)
        pose = f32(pose).reshape(12)
        ids = np.zeros((cap, 3), np.int32)
        new = np.zeros(cap, np.uint8)
        n = self.L.tfo_prepare(self.h, _p(depth, C.c_float), _p(pose, C.c_float),
                               _p(ids, C.c_int32), _p(new, C.c_uint8), cap)
        if n < 0:
            raise RuntimeError("oracle prepare: list capacity exceeded (%d)" % -n)
        return ids[:n].copy(), new[:n].copy()

    def integrate(self, depth, rgba, quality, pose, ids, needs, flag=1, kf_id=-1):
        depth = f32(depth)
        pose = f32(pose).reshape(12)
        ids = np.ascontiguousarray(ids, np.int32)
        n = len(ids)
        qout = np.zeros(max(n, 1), np.float32)
        r = self.L.tfo_integrate(self.h, _p(depth, C.c_float), _p(rgba, C.c_uint8),
                                 _p(quality, C.c_float), _p(pose, C.c_float), _p(ids, C.c_int32), n,
                                 int(flag), int(kf_id), _p(needs, C.c_uint8), _p(qout, C.c_float))
        if r != 0:
            raise RuntimeError("oracle integrate: %d chunks missing" % -r)
        return qout[:n]

    # ---- meshing ----
    def mesh_chunk(self, cid):
        """GenerateMeshEfficient of one chunk -> (verts [nv,3], normals, colors, indices u32[ni]) or None."""
        cid = np.ascontiguousarray(cid, np.int32)
        v = np.zeros((2187, 3), np.float32); nrm = np.zeros((2187, 3), np.float32)
        col = np.zeros((2187, 3), np.float32); idx = np.zeros(7680, np.uint32)
        ni = C.c_int64(0)
        nv = self.L.tfo_mesh_chunk(self.h, _p(cid, C.c_int32), _p(v, C.c_float), _p(nrm, C.c_float),
                                   _p(col, C.c_float), _p(idx, C.c_uint32), C.byref(ni))
        if nv < 0:
            return None
        return v[:nv].copy(), nrm[:nv].copy(), col[:nv].copy(), idx[:ni.value].copy()

    def update_meshes(self):
        return int(self.L.tfo_update_meshes(self.h))

    def list_meshes(self):
        n = int(self.L.tfo_volume_num_meshes(self.h))
        ids = np.zeros((max(n, 1), 3), np.int32)
        self.L.tfo_volume_list_meshes(self.h, _p(ids, C.c_int32), n)
        return ids[:n]

    def get_mesh(self, cid):
        """-> dict(verts, normals, colors, indices, adj u8[6], simplified) or None."""
        cid = np.ascontiguousarray(cid, np.int32)
        nv = C.c_int64(0); ni = C.c_int64(0)
        if self.L.tfo_volume_get_mesh(self.h, _p(cid, C.c_int32), C.byref(nv), C.byref(ni), None, None, None,
                                      None, None, None) != 0:
            return None
        v = np.zeros((max(nv.value, 1), 3), np.float32); nrm = np.zeros_like(v); col = np.zeros_like(v)
        idx = np.zeros(max(ni.value, 1), np.uint32)
        adj = np.zeros(6, np.uint8); simp = C.c_int(0)
        self.L.tfo_volume_get_mesh(self.h, _p(cid, C.c_int32), None, None, _p(v, C.c_float), _p(nrm, C.c_float),
                                   _p(col, C.c_float), _p(idx, C.c_uint32), _p(adj, C.c_uint8), C.byref(simp))
        return dict(verts=v[:nv.value], normals=nrm[:nv.value], colors=col[:nv.value], indices=idx[:ni.value],
                    adj=adj, simplified=bool(simp.value))

    def compress_meshes(self):
        """CompressMeshes on meshesToUpdate (cleared); returns chunksToUpdate in ascending id order."""
        n = int(self.L.tfo_volume_num_dirty(self.h))
        ids = np.zeros((max(n, 1), 3), np.int32)
        m = self.L.tfo_compress_meshes(self.h, _p(ids, C.c_int32), n)
        return ids[:m].copy()

    # ---- atlas stage on the volume's meshes ----
    def generate_patches(self, atlas, ids, labels, keyframes):
        """Chisel::GeneratePatches.  keyframes: dict kf_id -> (rgb u8[H,W,3], depth f32[H,W], T16); the arrays are
        kept alive by this object (Patch::image is a view).  Returns (rc, hot)."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        n = len(ids)
        order = sorted(keyframes)
        kfs = (Keyframe * max(len(order), 1))()
        self._kf_keep = getattr(self, "_kf_keep", {})
        for j, k in enumerate(order):
            rgb, depth, T = keyframes[k]
            rgb = np.ascontiguousarray(rgb, np.uint8); depth = f32(depth)
            self._kf_keep[k] = (rgb, depth)
            kfs[j].rgb = _p(rgb, C.c_uint8); kfs[j].depth = _p(depth, C.c_float); kfs[j].kf_id = int(k)
            for q, val in enumerate(f32(T).reshape(16)):
                kfs[j].T[q] = val
        idx = np.array([order.index(int(l)) for l in labels], np.int32)
        hot = np.zeros(2, np.uint64)
        rc = self.L.tfo_generate_patches(self.h, atlas.h, _p(ids, C.c_int32), n, _p(idx, C.c_int32), kfs,
                                         _p(hot, C.c_uint64))
        return rc, (int(hot[0]), int(hot[1]))

    def update_atlas(self, atlas, ids):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        self.L.tfo_update_atlas(self.h, atlas.h, _p(ids, C.c_int32), len(ids))

    def compensate_color(self):
        return int(self.L.tfo_compensate_color_volume(self.h))

    def draw_meshes(self, atlas):
        ni = C.c_int64(0)
        nv = self.L.tfo_draw_meshes(self.h, atlas.h, None, None, 0, 0, C.byref(ni))
        V = np.zeros((max(nv, 1), 12), np.float32); I = np.zeros(max(ni.value, 1), np.uint32)
        self.L.tfo_draw_meshes(self.h, atlas.h, _p(V, C.c_float), _p(I, C.c_uint32), nv, ni.value, C.byref(ni))
        return V[:nv], I[:ni.value]

    def get_patch(self, cid):
        cid = np.ascontiguousarray(cid, np.int32)
        tl = C.c_uint64(0); fid = C.c_int(0); flags = C.c_int(0); pnv = C.c_int64(0)
        bbox = np.zeros(4, np.int32); ratio = np.zeros(2, np.float32)
        if self.L.tfo_volume_get_patch(self.h, _p(cid, C.c_int32), C.byref(tl), C.byref(fid), _p(bbox, C.c_int32),
                                       C.byref(flags), _p(ratio, C.c_float), C.byref(pnv), None, None, None) != 0:
            return None
        n = pnv.value
        tc = np.zeros((max(n, 1), 2), np.float32); tcol = np.zeros((max(n, 1), 3), np.float32)
        labs = np.zeros((max(n, 1), 3), np.float32)
        self.L.tfo_volume_get_patch(self.h, _p(cid, C.c_int32), None, None, None, None, None, None,
                                    _p(tc, C.c_float), _p(tcol, C.c_float), _p(labs, C.c_float))
        return dict(texloc=int(tl.value), frameid=fid.value, bbox=bbox, flags=flags.value, ratio=ratio,
                    texcoord=tc[:n], texcolor=tcol[:n], labs=labs[:n])

    def frame_textured(self, atlas, depth, rgba, pose, pose_inv16, frame_id):
        """The textured per-frame unit; returns the number of patches (chunksToUpdate)."""
        depth = f32(depth); rgba = np.ascontiguousarray(rgba, np.uint8)
        pose = f32(pose).reshape(12); T = f32(pose_inv16).reshape(16)
        self._scratch = getattr(self, "_scratch", None)
        if self._scratch is None or self._scratch.size != depth.size * 3:
            self._scratch = np.zeros(depth.size * 3, np.uint8)
        self._frame_keep = (depth, rgba)
        return int(self.L.tfo_frame_textured(self.h, atlas.h, _p(depth, C.c_float), _p(rgba, C.c_uint8),
                                             _p(pose, C.c_float), _p(T, C.c_float), int(frame_id),
                                             _p(self._scratch, C.c_uint8)))

    def finalize(self, ids, needs, new):
        ids = np.ascontiguousarray(ids, np.int32)
        n = len(ids)
        valid = np.zeros((max(n, 1), 3), np.int32)
        nv = self.L.tfo_finalize(self.h, _p(ids, C.c_int32), _p(needs, C.c_uint8),
                                 _p(new, C.c_uint8), n, _p(valid, C.c_int32))
        return valid[:nv].copy()

    def integrate_frame(self, depth, rgba, pose):
        depth = f32(depth)
        pose = f32(pose).reshape(12)
        ns = C.c_int64(0)
        nv = self.L.tfo_integrate_frame(self.h, _p(depth, C.c_float), _p(rgba, C.c_uint8),
                                        _p(pose, C.c_float), C.byref(ns))
        return int(nv), int(ns.value)


class Atlas:
    def __init__(self, res, w=0, h=0):
        self.L = lib()
        self.h = self.L.tfo_atlas_create(np.float32(res), w, h)
        self.w = w or 13824
        self.hh = h or 13824
        self.pw = self.L.tfo_atlas_patch_w(self.h)
        self.ph = self.L.tfo_atlas_patch_h(self.h)

    def close(self):
        if self.h:
            self.L.tfo_atlas_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self):
        t = C.c_uint64(0)
        r = self.L.tfo_atlas_alloc(self.h, C.byref(t))
        return r, int(t.value)

    def loc_next(self):
        return int(self.L.tfo_atlas_loc_next(self.h))

    def buffer(self):
        p = self.L.tfo_atlas_buffer(self.h)
        return np.ctypeslib.as_array(p, shape=(self.hh, self.w, 3))

    def blit(self, texloc, rgb, bbox, ratio=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        bbox = np.ascontiguousarray(bbox, np.int32)
        ratio = np.ones(2, np.float32) if ratio is None else f32(ratio)
        r = self.L.tfo_atlas_blit(self.h, texloc, _p(rgb, C.c_uint8), rgb.shape[1], rgb.shape[0],
                                  _p(bbox, C.c_int32), _p(ratio, C.c_float))
        return r, ratio

    def patches_batch(self, texlocs, voff, verts, cols, T16, rgb, depth, cam):
        """GeneratePatches + UpdateAtlas for the patches of one keyframe in one C call (CPU baseline)."""
        tl = np.ascontiguousarray(texlocs, np.uint64)
        voff = np.ascontiguousarray(voff, np.int64)
        verts = np.ascontiguousarray(verts, np.float32); cols = np.ascontiguousarray(cols, np.float32)
        T16 = np.ascontiguousarray(T16, np.float32)
        nv = int(voff[-1])
        tc = np.zeros((max(nv, 1), 2), np.float32); tcol = np.zeros((max(nv, 1), 3), np.float32)
        self.L.tfo_patches_batch(self.h, len(tl), _p(tl, C.c_uint64), _p(voff, C.c_int64), _p(verts, C.c_float),
                                 _p(cols, C.c_float), _p(T16, C.c_float), _p(rgb, C.c_uint8), _p(depth, C.c_float),
                                 C.byref(cam), _p(tc, C.c_float), _p(tcol, C.c_float))
        return tc[:nv], tcol[:nv]

    def hot_range(self, texlocs):
        t = np.ascontiguousarray(texlocs, np.uint64)
        a = C.c_uint64(0)
        b = C.c_uint64(0)
        self.L.tfo_atlas_hot_range(self.h, _p(t, C.c_uint64), len(t), C.byref(a), C.byref(b))
        return int(a.value), int(b.value)


def patch_project(verts, colors, T, rgb, depth, cam: Camera):
    verts = f32(verts).reshape(-1, 3)
    colors = f32(colors).reshape(-1, 3)
    n = len(verts)
    T = f32(T).reshape(16)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    depth = f32(depth)
    tc = np.zeros((max(n, 1), 2), np.float32)
    tcol = np.zeros((max(n, 1), 3), np.float32)
    bb = np.zeros(4, np.int32)
    wm = C.c_int(0)
    nc = C.c_int64(0)
    flag = lib().tfo_patch_project(_p(verts, C.c_float), _p(colors, C.c_float), n, _p(T, C.c_float),
                                   _p(rgb, C.c_uint8), _p(depth, C.c_float), C.byref(cam),
                                   _p(tc, C.c_float), _p(tcol, C.c_float), _p(bb, C.c_int32),
                                   C.byref(wm), C.byref(nc))
    return dict(flag=flag, texcoord=tc[:n], texcolor=tcol[:n], bbox=bb, wrong_mapping=bool(wm.value),
                n_caution=int(nc.value))


def color_transfer(cov_src, cov_tar):
    """Transfer matrix of Chisel::CompensateColor (Chisel.cpp:247-266)."""
    a = np.ascontiguousarray(cov_src, np.float32).reshape(9)
    b = np.ascontiguousarray(cov_tar, np.float32).reshape(9)
    T = np.zeros(9, np.float32)
    lib().tfo_color_transfer(_p(a, C.c_float), _p(b, C.c_float), _p(T, C.c_float))
    return T.reshape(3, 3)


def color_compensate(frame_ids, wrong_mapping, has_adjusted, voff, texcolor, meshcolor):
    """Chisel::CompensateColor over a batch of patches.  Returns (labs, has_adjusted, T per cluster, cluster per patch)."""
    frame_ids = np.ascontiguousarray(frame_ids, np.int32)
    n = len(frame_ids)
    wrong = np.ascontiguousarray(wrong_mapping, np.uint8)
    adj = np.ascontiguousarray(has_adjusted, np.uint8).copy()
    voff = np.ascontiguousarray(voff, np.int64)
    tc = np.ascontiguousarray(texcolor, np.float32).reshape(-1, 3)
    mc = np.ascontiguousarray(meshcolor, np.float32).reshape(-1, 3)
    labs = np.full_like(tc, np.nan)
    T = np.zeros((max(n, 1), 9), np.float32)
    cl = np.full(max(n, 1), -1, np.int32)
    ncl = lib().tfo_color_compensate(n, _p(frame_ids, C.c_int32), _p(wrong, C.c_uint8), _p(adj, C.c_uint8),
                                     _p(voff, C.c_int64), _p(tc, C.c_float), _p(mc, C.c_float), _p(labs, C.c_float),
                                     _p(T, C.c_float), _p(cl, C.c_int32))
    return labs, adj, T[:ncl].reshape(-1, 3, 3), cl[:n]


def pack_vertices(complete, wrong_mapping, labs_valid, texloc, ratio, atlas_w, atlas_h, voff, verts, colors, normals,
                  texcoord, texcolor, labs, ioff, indices):
    """Chisel::DrawMeshes (Chisel.cpp:288-355) -> (vertices f32[n,12], indices u32[m])."""
    a8 = lambda x: np.ascontiguousarray(x, np.uint8)
    f = lambda x: np.ascontiguousarray(x, np.float32)
    complete, wrong_mapping, labs_valid = a8(complete), a8(wrong_mapping), a8(labs_valid)
    texloc = np.ascontiguousarray(texloc, np.uint64)
    voff = np.ascontiguousarray(voff, np.int64); ioff = np.ascontiguousarray(ioff, np.int64)
    indices = np.ascontiguousarray(indices, np.uint32)
    ratio, verts, colors, normals, texcoord, texcolor, labs = map(f, (ratio, verts, colors, normals, texcoord, texcolor, labs))
    out_v = np.zeros((max(int(voff[-1]), 1), 12), np.float32)
    out_i = np.zeros(max(int(ioff[-1]), 1), np.uint32)
    ni = C.c_int64(0)
    nv = lib().tfo_pack_vertices(len(complete), _p(complete, C.c_uint8), _p(wrong_mapping, C.c_uint8),
                                 _p(labs_valid, C.c_uint8), _p(texloc, C.c_uint64), _p(ratio, C.c_float),
                                 int(atlas_w), int(atlas_h), _p(voff, C.c_int64), _p(verts, C.c_float),
                                 _p(colors, C.c_float), _p(normals, C.c_float), _p(texcoord, C.c_float),
                                 _p(texcolor, C.c_float), _p(labs, C.c_float), _p(ioff, C.c_int64),
                                 _p(indices, C.c_uint32), _p(out_v, C.c_float), _p(out_i, C.c_uint32), C.byref(ni))
    return out_v[:nv], out_i[:ni.value]


# ---- frame pre-processing (BasicAPI.cpp:378-905): numpy in, numpy out ------------------------------------
def _k4(cam):
    return [C.c_float(cam.fx), C.c_float(cam.fy), C.c_float(cam.cx), C.c_float(cam.cy)]


def pre_normal_map(depth, cam):
    depth = f32(depth)
    H, W = depth.shape
    n = np.zeros((3, H, W), np.float32)
    lib().tfo_pre_normal_map(_p(depth, C.c_float), W, H, *_k4(cam), _p(n, C.c_float))
    return n


def pre_refine_depth_normal(normal, depth, cam):
    """-> (normal, depth) refined copies"""
    n, d = f32(normal).copy(), f32(depth).copy()
    H, W = d.shape
    lib().tfo_pre_refine_depth_normal(_p(n, C.c_float), _p(d, C.c_float), W, H, *_k4(cam))
    return n, d


def pre_color_valid(normal, cam):
    n = f32(normal)
    _, H, W = n.shape
    flag = np.zeros((H, W), np.uint8)
    lib().tfo_pre_color_valid(_p(n, C.c_float), W, H, *_k4(cam), _p(flag, C.c_uint8))
    return flag


def pre_color_quality(depth, normal, rgb, cam):
    d, n = f32(depth), f32(normal)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W = d.shape
    q = np.zeros((H, W), np.float32)
    lib().tfo_pre_color_quality(_p(d, C.c_float), _p(n, C.c_float), _p(rgb, C.c_uint8), W, H, *_k4(cam), _p(q, C.c_float))
    return q


def pre_refine_newframe(depth_ref, depth_new, cam, T12):
    r, d = f32(depth_ref), f32(depth_new).copy()
    T = f32(T12).reshape(12)
    H, W = d.shape
    lib().tfo_pre_refine_newframe(_p(r, C.c_float), _p(d, C.c_float), W, H, *_k4(cam), _p(T, C.c_float))
    return d


def pre_refine_keyframe(depth_ref, weight_ref, depth_new, cam, T12):
    """-> (depth_ref, weight_ref) refined copies"""
    r, w, d = f32(depth_ref).copy(), f32(weight_ref).copy(), f32(depth_new)
    T = f32(T12).reshape(12)
    H, W = r.shape
    lib().tfo_pre_refine_keyframe(_p(r, C.c_float), _p(w, C.c_float), _p(d, C.c_float), W, H, *_k4(cam), _p(T, C.c_float))
    return r, w


def pre_frame_depth(depth_u16, maximum_depth, depth_scale, d=9, sigma_color=0.03, sigma_space=10.0):
    """DatasetWrapper::framePreprocess -> (depth u16 after the write-back, refined depth f32)"""
    dz = np.ascontiguousarray(depth_u16, np.uint16).copy()
    H, W = dz.shape
    out = np.zeros((H, W), np.float32)
    lib().tfo_pre_frame_depth(_p(dz, C.c_uint16), W, H, C.c_float(maximum_depth), C.c_float(depth_scale), int(d),
                              C.c_double(sigma_color), C.c_double(sigma_space), _p(out, C.c_float))
    return dz, out
