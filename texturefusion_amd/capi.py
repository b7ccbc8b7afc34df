"""ctypes binding of include/tf_fusion.h (texturefusion_amd/libtexfusion_hip.so).

Plumbing only: every compute call goes through the C ABI into the hand-written gfx950 kernels.
There is no fallback path -- a missing library raises ImportError, a missing GPU makes
``Volume()`` raise :class:`TFError` (TF_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TF_LIB points at another build of the same sources (A/B runs of tuning variants under variants/)
LIB_PATH = os.environ.get("TF_LIB") or os.path.join(_HERE, "libtexfusion_hip.so")

TF_OK = 0
TF_ERR_ATLAS_FULL = -1
TF_ERR_INVALID = -2
TF_ERR_CAPACITY = -3
TF_ERR_HIP = -4
TF_ERR_NO_DEVICE = -5
TF_ERR_MISSING_CHUNK = -6
TF_BOUNDARY_RECORD_BYTES = 16 + 4096 + 4096

PROF_NAMES = ("bbox", "select", "scan", "emit", "integrate", "finalize", "patch_project",
              "atlas_blit", "mesh", "dirty", "patch_rank", "xchg", "xchg_wait")

# every symbol include/tf_fusion.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "tf_last_error", "tf_device_count", "tf_volume_create", "tf_volume_create_sized", "tf_volume_destroy", "tf_volume_reset",
    "tf_set_stream", "tf_set_camera", "tf_set_truncation", "tf_set_weight", "tf_frame_upload",
    "tf_frame_upload_rgb",
    "tf_frame_bind_device", "tf_prepare", "tf_integrate", "tf_finalize", "tf_integrate_frame",
    "tf_integrate_frames_device", "tf_sync", "tf_has_chunk", "tf_chunk_download",
    "tf_chunks_download", "tf_chunk_upload", "tf_list_chunks", "tf_list_dirty", "tf_clear_dirty",
    "tf_get_stats", "tf_profile_enable", "tf_profile_get", "tf_profile_calibrate", "tf_keyframe_unit_device", "tf_keyframe_unit_release", "tf_keyframe_unit_stats", "tf_keyframe_unit_stats_ex", "tf_observations_record", "tf_observations_retract", "tf_export_datacost", "tf_export_adjacency", "tf_debug_phase_raw", "tf_set_partition", "tf_set_partition_key", "tf_boundary_pack", "tf_boundary_pack_async",
    "tf_boundary_unpack", "tf_keyframe_cache", "tf_keyframe_cache_device", "tf_keyframe_set_pose",
    "tf_keyframe_release", "tf_atlas_patch_size", "tf_atlas_loc_next", "tf_atlas_size", "tf_meshes_upload",
    "tf_generate_patches", "tf_compensate_color", "tf_update_atlas", "tf_draw_meshes", "tf_draw_meshes_device",
    "tf_patches_download", "tf_atlas_download_rows", "tf_atlas_snapshot_rows", "tf_stream_frames_device",
    "tf_stream_frames_textured_device", "tf_get_texture_stats", "tf_integrate_frame_host", "tf_integrate_frame_host_rgb", "tf_host_frame_times", "tf_host_register", "tf_host_unregister",
    "tf_host_frame_buffers", "tf_host_frame_deferral", "tf_host_frame_set_deferral", "tf_host_frame_set_async", "tf_host_frame_fence", "tf_texture_frame_device_phase", "tf_comm_exchange_overlap", "tf_texture_frame_device", "tf_boundary_block_bytes", "tf_boundary_pack_block", "tf_boundary_pack_bands", "tf_boundary_band_bounds", "tf_boundary_pack_bands2", "tf_boundary_unpack_pair", "tf_comm_exchange_mode", "tf_comm_stats", "tf_comm_stats_ex",
    "tf_boundary_unpack_blocks", "tf_comm_unique_id", "tf_comm_init", "tf_comm_destroy", "tf_exchange_boundary",
    "tf_comm_exchange_every_frame",
    "tf_update_meshes", "tf_check_summaries", "tf_check_neighbours", "tf_list_meshes", "tf_mesh_counts", "tf_meshes_download", "tf_compress_meshes",
    "tf_pre_normal_map", "tf_pre_refine_depth_normal", "tf_pre_color_valid", "tf_pre_color_quality",
    "tf_pre_refine_newframe", "tf_pre_refine_keyframe", "tf_pre_frame_depth", "tf_integrate_depth_group", "tf_integrate_depth_group_host",
)


def host_frame_deferral():
    """(frames tf_integrate_frame_host runs behind its caller, staging slots of its ring) -- needs no GPU"""
    a, b = C.c_int32(0), C.c_int32(0)
    lib().tf_host_frame_deferral(None, C.byref(a), C.byref(b))
    return a.value, b.value


class TFError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tf error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_chunks", C.c_int64), ("max_list", C.c_int64),
                ("max_coarse", C.c_int64), ("atlas_w", C.c_int32), ("atlas_h", C.c_int32),
                ("max_keyframes", C.c_int32), ("mesh_overflow_blocks", C.c_int32),
                ("mesh_max_vertices", C.c_int32), ("mesh_max_triangles", C.c_int32), ("mesh_blocks", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("n_coarse", C.c_int64), ("n_selected", C.c_int64), ("n_updated", C.c_int64),
                ("rows_tsdf", C.c_int64), ("rows_color", C.c_int64), ("n_chunks", C.c_int64),
                ("n_slots", C.c_int64), ("n_dirty", C.c_int64), ("min_id", C.c_int32 * 3),
                ("max_id", C.c_int32 * 3), ("n_listed", C.c_int64)]


class TextureStats(C.Structure):
    _fields_ = [("n_dirty", C.c_int64), ("n_meshes", C.c_int64), ("n_vertices", C.c_int64),
                ("n_triangles", C.c_int64), ("roi_pixels", C.c_int64), ("n_patches", C.c_int64),
                ("n_slots", C.c_int64), ("n_exact", C.c_int64), ("n_survivors", C.c_int64),
                ("n_surface", C.c_int64)]


class UnitFrame(C.Structure):
    _fields_ = [("d_depth", C.c_void_p), ("d_rgba", C.c_void_p), ("d_quality", C.c_void_p), ("pose", C.c_float * 12)]


class UnitGroup(C.Structure):
    _fields_ = [("kf_id", C.c_int32), ("n_local", C.c_int32), ("keyframe", UnitFrame), ("local", UnitFrame * 6),
                ("old_keyframe_pose", C.c_float * 12), ("old_local_pose", (C.c_float * 12) * 6)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * len(PROF_NAMES)), ("launches", C.c_int64 * len(PROF_NAMES))]  # TF_PROF_COUNT


_lib = None


def lib():
    """Load the HIP library (ImportError if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "texturefusion_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, fp, u8p, u16p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
    i32p, i64p, u64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    L.tf_last_error.restype = C.c_char_p
    L.tf_device_count.restype = C.c_int
    L.tf_volume_create.argtypes = [i32p, C.c_float, C.c_int, C.POINTER(Config), C.POINTER(vp)]
    L.tf_volume_destroy.argtypes = [vp]
    L.tf_volume_reset.argtypes = [vp]
    L.tf_set_stream.argtypes = [vp, vp]
    L.tf_set_camera.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                C.c_float, C.c_float]
    L.tf_set_truncation.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    L.tf_set_weight.argtypes = [vp, C.c_float]
    L.tf_frame_upload.argtypes = [vp, fp, u8p, fp]
    L.tf_frame_upload_rgb.argtypes = [vp, fp, u8p, u8p, fp]
    L.tf_frame_bind_device.argtypes = [vp, vp, vp, vp]
    L.tf_prepare.argtypes = [vp, fp, i32p, u8p, C.c_int64, i64p]
    L.tf_integrate.argtypes = [vp, fp, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, u8p, fp]
    L.tf_finalize.argtypes = [vp, i32p, u8p, u8p, C.c_int64, i32p, i64p]
    L.tf_integrate_frame.argtypes = [vp, fp, C.c_int]
    L.tf_integrate_frames_device.argtypes = [vp, C.c_int64, C.POINTER(vp), C.POINTER(vp), fp]
    L.tf_sync.argtypes = [vp]
    L.tf_has_chunk.argtypes = [vp, i32p, C.POINTER(C.c_int)]
    L.tf_chunk_download.argtypes = [vp, i32p, fp, fp, u16p]
    L.tf_chunks_download.argtypes = [vp, i32p, C.c_int64, fp, fp, u16p]
    L.tf_chunk_upload.argtypes = [vp, i32p, fp, fp, u16p]
    L.tf_list_chunks.argtypes = [vp, i32p, C.c_int64, i64p]
    L.tf_list_dirty.argtypes = [vp, i32p, C.c_int64, i64p]
    L.tf_clear_dirty.argtypes = [vp]
    L.tf_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.tf_profile_enable.argtypes = [vp, C.c_uint32]
    L.tf_profile_get.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.tf_profile_calibrate.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    L.tf_keyframe_unit_device.argtypes = [vp, C.POINTER(UnitGroup), C.POINTER(UnitGroup), C.c_int32, C.c_int32, fp]
    L.tf_keyframe_unit_release.argtypes = [vp]
    L.tf_keyframe_unit_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tf_keyframe_unit_stats_ex.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tf_observations_record.argtypes = [vp, C.c_int32]
    L.tf_observations_retract.argtypes = [vp, C.c_int32, i32p, C.c_int64]
    L.tf_export_datacost.argtypes = [vp, i32p, C.c_int64, C.c_int32, i32p, C.c_int32, fp]
    L.tf_export_adjacency.argtypes = [vp, i32p, C.c_int64, i32p, C.c_int64, i64p]
    L.tf_debug_phase_raw.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int64]
    L.tf_set_partition.argtypes = [vp, C.c_int32, C.c_int32]
    L.tf_set_partition_key.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.tf_boundary_pack.argtypes = [vp, vp, C.c_int64, i64p]
    L.tf_boundary_pack_async.argtypes = [vp, vp, C.c_int64, vp]
    L.tf_boundary_unpack.argtypes = [vp, vp, C.c_int64]
    L.tf_keyframe_cache.argtypes = [vp, C.c_int32, u8p, fp]
    L.tf_keyframe_cache_device.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
    L.tf_keyframe_set_pose.argtypes = [vp, C.c_int32, fp]
    L.tf_keyframe_release.argtypes = [vp, C.c_int32]
    L.tf_atlas_patch_size.argtypes = [vp, i32p, i32p]
    L.tf_atlas_size.argtypes = [vp, i32p, i32p]
    L.tf_atlas_loc_next.argtypes = [vp, u64p]
    L.tf_meshes_upload.argtypes = [vp, i32p, C.c_int64, i64p, i64p, fp, fp, fp, C.POINTER(C.c_uint32)]
    L.tf_generate_patches.argtypes = [vp, i32p, C.c_int64, i32p, u64p]
    L.tf_compensate_color.argtypes = [vp, i64p]
    L.tf_update_atlas.argtypes = [vp, i32p, C.c_int64]
    L.tf_draw_meshes.argtypes = [vp, fp, C.POINTER(C.c_uint32), C.c_int64, C.c_int64, i64p, i64p]
    L.tf_draw_meshes_device.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, i64p, i64p]
    L.tf_patches_download.argtypes = [vp, i32p, C.c_int64, i64p, u64p, i32p, i32p, i32p, fp, fp, fp, fp]
    L.tf_stream_frames_device.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(vp), C.POINTER(vp), fp]
    L.tf_stream_frames_textured_device.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(vp), C.POINTER(vp), fp, fp,
                                                   C.c_int32]
    L.tf_get_texture_stats.argtypes = [vp, C.POINTER(TextureStats)]
    L.tf_integrate_frame_host.argtypes = [vp, fp, u8p, fp, fp, C.c_int32]
    L.tf_integrate_frame_host_rgb.argtypes = [vp, fp, u8p, u8p, fp, fp, C.c_int32]
    L.tf_host_frame_times.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.tf_host_register.argtypes = [vp, C.c_void_p, C.c_int64]
    L.tf_host_unregister.argtypes = [vp, C.c_void_p]
    L.tf_host_frame_buffers.argtypes = [vp, C.POINTER(fp), C.POINTER(u8p)]
    L.tf_host_frame_deferral.argtypes = [vp, i32p, i32p]
    L.tf_host_frame_set_deferral.argtypes = [vp, C.c_int]
    L.tf_host_frame_set_async.argtypes = [vp, C.c_int]
    L.tf_host_frame_fence.argtypes = [vp]
    L.tf_texture_frame_device.argtypes = [vp, fp, C.c_int32]
    L.tf_texture_frame_device_phase.argtypes = [vp, fp, C.c_int32, C.c_int]
    L.tf_comm_exchange_overlap.argtypes = [vp, C.c_int]
    L.tf_boundary_block_bytes.restype = C.c_size_t
    L.tf_boundary_block_bytes.argtypes = [C.c_int64]
    L.tf_boundary_pack_block.argtypes = [vp, vp, C.c_int64]
    L.tf_boundary_pack_bands.argtypes = [vp, vp, vp, C.c_int64]
    L.tf_boundary_band_bounds.argtypes = [vp, C.c_int64, i64p]
    L.tf_boundary_pack_bands2.argtypes = [vp, vp, C.c_int64, vp, C.c_int64]
    L.tf_boundary_unpack_pair.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int]
    L.tf_comm_stats_ex.argtypes = [vp, i64p]
    L.tf_comm_exchange_mode.argtypes = [vp, C.c_int]
    L.tf_comm_stats.argtypes = [vp, i64p, i64p]
    L.tf_boundary_unpack_blocks.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int]
    L.tf_comm_unique_id.argtypes = [vp]
    L.tf_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.tf_comm_destroy.argtypes = [vp]
    L.tf_exchange_boundary.argtypes = [vp, C.c_int64]
    L.tf_integrate_depth_group.argtypes = [vp, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                           C.c_int64, C.c_int, C.POINTER(C.c_uint8)]
    L.tf_integrate_depth_group_host.argtypes = L.tf_integrate_depth_group.argtypes
    L.tf_pre_normal_map.argtypes = [vp, vp, vp]
    L.tf_pre_refine_depth_normal.argtypes = [vp, vp, vp]
    L.tf_pre_color_valid.argtypes = [vp, vp, vp]
    L.tf_pre_color_quality.argtypes = [vp, vp, vp, vp, vp]
    L.tf_pre_refine_newframe.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    L.tf_pre_frame_depth.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_int, C.c_double, C.c_double]
    L.tf_pre_refine_keyframe.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.tf_comm_exchange_every_frame.argtypes = [vp, C.c_int64]
    L.tf_atlas_download_rows.argtypes = [vp, C.c_int64, C.c_int64, u8p]
    L.tf_atlas_snapshot_rows.argtypes = [vp, C.c_int64, C.c_int64, u8p, i64p, i32p]
    u32p = C.POINTER(C.c_uint32)
    L.tf_update_meshes.argtypes = [vp, i64p]
    L.tf_list_meshes.argtypes = [vp, i32p, C.c_int64, i64p]
    L.tf_check_summaries.argtypes = [vp, i64p, i64p, i64p]
    L.tf_check_neighbours.argtypes = [vp, i64p]
    L.tf_mesh_counts.argtypes = [vp, i32p, C.c_int64, i32p, i32p, u8p, u8p]
    L.tf_meshes_download.argtypes = [vp, i32p, C.c_int64, i64p, i64p, fp, fp, fp, u32p]
    L.tf_compress_meshes.argtypes = [vp, i32p, C.c_int64, i64p]
    _lib = L
    return L


def boundary_block_bytes(cap):
    return int(lib().tf_boundary_block_bytes(cap))


def comm_unique_id():
    """128-byte RCCL id (ncclGetUniqueId) for tf_comm_init."""
    buf = (C.c_uint8 * 128)()
    rc = lib().tf_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != TF_OK:
        raise TFError(rc, lib().tf_last_error().decode())
    return bytes(buf)


def _p(a, ty):
    """pointer to a numpy array's data.  Not ndarray.ctypes.data_as: that builds a reference cycle per call, and the
    collections it triggers cost 30-60 us per pointer in a process with many live objects (e.g. after `import
    torch`) -- more than the whole per-frame call."""
    if a is None:
        return None
    ptr = C.cast(a.__array_interface__["data"][0], C.POINTER(ty))
    ptr._keep = a  # the array outlives the pointer (plain reference, no cycle)
    return ptr


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Volume:
    """Device-resident chunk volume + atlas behind one tf_volume handle."""

    def __init__(self, res, cam=None, max_chunks=1 << 17, max_list=1 << 18, max_coarse=1 << 20,
                 atlas_w=0, atlas_h=0, device=0, use_color=True, stream=None, mesh_max_vertices=0,
                 mesh_max_triangles=0, mesh_overflow_blocks=0, mesh_blocks=0):
        self.L = lib()
        self.h = C.c_void_p()
        cfg = Config(device, max_chunks, max_list, max_coarse, atlas_w, atlas_h, 0, mesh_overflow_blocks, mesh_max_vertices,
                     mesh_max_triangles, mesh_blocks)
        dims = (C.c_int32 * 3)(8, 8, 8)
        rc = self.L.tf_volume_create(dims, np.float32(res), int(use_color), C.byref(cfg), C.byref(self.h))
        if rc != TF_OK:
            self.h = None
            raise TFError(rc, self.L.tf_last_error().decode())
        self.res = np.float32(res)
        if stream is not None:
            self._ck(self.L.tf_set_stream(self.h, C.c_void_p(stream)))
        if cam is not None:
            self.set_camera(cam)

    def _ck(self, rc):
        if rc != TF_OK:
            raise TFError(rc, self.L.tf_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.tf_volume_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters
    def set_camera(self, cam):
        self.cam = cam
        self._ck(self.L.tf_set_camera(self.h, cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height,
                                      cam.near, cam.far))

    def set_truncation(self, q, l, c, s):
        self._ck(self.L.tf_set_truncation(self.h, q, l, c, s))

    def set_weight(self, w):
        self._ck(self.L.tf_set_weight(self.h, w))

    def reset(self):
        self._ck(self.L.tf_volume_reset(self.h))

    def set_partition(self, lo, hi, axis=(1, 0, 0)):
        """Own the chunks with lo <= axis . id < hi (axis in {0,1}^3; (1,0,0) = ChunkID.x slabs)."""
        self._ck(self.L.tf_set_partition_key(self.h, int(axis[0]), int(axis[1]), int(axis[2]), lo, hi))

    # -- frames
    def frame_upload(self, depth, rgba=None, quality=None):
        depth = _f32(depth)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, np.uint8)
        quality = None if quality is None else _f32(quality)
        self._check_images(depth, rgba)
        self._check_images(quality)
        self._keep = (depth, rgba, quality)
        self._ck(self.L.tf_frame_upload(self.h, _p(depth, C.c_float), _p(rgba, C.c_uint8),
                                        _p(quality, C.c_float)))

    def frame_upload_rgb(self, depth, rgb, color_valid, quality=None):
        """Frame::rgb (u8[H,W,3]) + Frame::colorValidFlag (u8[H,W]) packed into the path's RGBA on the device."""
        depth = _f32(depth)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        color_valid = np.ascontiguousarray(color_valid, np.uint8)
        quality = None if quality is None else _f32(quality)
        self._keep = (depth, rgb, color_valid, quality)
        self._ck(self.L.tf_frame_upload_rgb(self.h, _p(depth, C.c_float), _p(rgb, C.c_uint8),
                                            _p(color_valid, C.c_uint8), _p(quality, C.c_float)))

    def frame_bind_device(self, d_depth, d_rgba=0, d_quality=0):
        self._ck(self.L.tf_frame_bind_device(self.h, C.c_void_p(d_depth), C.c_void_p(d_rgba or None),
                                             C.c_void_p(d_quality or None)))

    # -- the reference's call-by-call flow
    def prepare(self, pose, cap=1 << 18):
        pose = _f32(pose).reshape(12)
        ids = np.zeros((cap, 3), np.int32)
        new = np.zeros(cap, np.uint8)
        n = C.c_int64(0)
        self._ck(self.L.tf_prepare(self.h, _p(pose, C.c_float), _p(ids, C.c_int32), _p(new, C.c_uint8),
                                   cap, C.byref(n)))
        return ids[:n.value].copy(), new[:n.value].copy()

    def integrate(self, pose, ids, needs, flag=1, use_color=True, use_quality=False):
        pose = _f32(pose).reshape(12)
        ids = np.ascontiguousarray(ids, np.int32)
        n = len(ids)
        q = np.zeros(max(n, 1), np.float32)
        self._ck(self.L.tf_integrate(self.h, _p(pose, C.c_float), _p(ids, C.c_int32), n, int(flag),
                                     int(use_color), int(use_quality), _p(needs, C.c_uint8),
                                     _p(q, C.c_float)))
        return q[:n]

    def finalize(self, ids, needs, new):
        ids = np.ascontiguousarray(ids, np.int32)
        n = len(ids)
        valid = np.zeros((max(n, 1), 3), np.int32)
        nv = C.c_int64(0)
        self._ck(self.L.tf_finalize(self.h, _p(ids, C.c_int32), _p(needs, C.c_uint8), _p(new, C.c_uint8),
                                    n, _p(valid, C.c_int32), C.byref(nv)))
        return valid[:nv.value].copy()

    # -- fused per-frame unit
    def integrate_frame(self, pose, use_color=True):
        pose = _f32(pose).reshape(12)
        self._ck(self.L.tf_integrate_frame(self.h, _p(pose, C.c_float), int(use_color)))

    def integrate_frames_device(self, d_depths, d_rgbas, poses):
        n = len(d_depths)
        poses = _f32(poses).reshape(n, 12)
        dd = (C.c_void_p * n)(*d_depths)
        dr = None if d_rgbas is None else (C.c_void_p * n)(*d_rgbas)
        self._ck(self.L.tf_integrate_frames_device(self.h, n, dd, dr, _p(poses, C.c_float)))

    def stream_frames_device(self, d_depths, d_rgbas, poses, n_ahead=0):
        """d_depths / d_rgbas / poses hold n + n_ahead frames: the first n are integrated, the rest only selected."""
        m = len(d_depths)
        poses = _f32(poses).reshape(m, 12)
        dd = (C.c_void_p * m)(*d_depths)
        dr = None if d_rgbas is None else (C.c_void_p * m)(*d_rgbas)
        self._ck(self.L.tf_stream_frames_device(self.h, m - n_ahead, n_ahead, dd, dr, _p(poses, C.c_float)))

    def stream_frames_textured_device(self, d_depths, d_rgbas, poses, pose_inv, first_frame_id, n_ahead=0):
        m = len(d_depths)
        poses = _f32(poses).reshape(m, 12)
        pose_inv = _f32(pose_inv).reshape(m, 16)
        dd = (C.c_void_p * m)(*d_depths)
        dr = (C.c_void_p * m)(*d_rgbas)
        self._ck(self.L.tf_stream_frames_textured_device(self.h, m - n_ahead, n_ahead, dd, dr, _p(poses, C.c_float),
                                                         _p(pose_inv, C.c_float), int(first_frame_id)))

    def integrate_frame_host(self, depth, rgba, pose, pose_inv16=None, frame_id=0):
        """MobileFusion::IntegrateFrame with host images (asynchronous; pose_inv16 = textured unit)."""
        depth = _f32(depth)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, np.uint8)
        self._check_images(depth, rgba)
        pose = _f32(pose).reshape(12)
        T = None if pose_inv16 is None else _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_integrate_frame_host(self.h, _p(depth, C.c_float), _p(rgba, C.c_uint8), _p(pose, C.c_float),
                                                _p(T, C.c_float), int(frame_id)))

    def integrate_frame_host_addr(self, depth_addr, rgba_addr, pose_addr, pose_inv16_addr, frame_id=0):
        """integrate_frame_host for a driver loop that has its arrays' ADDRESSES at hand (array.ctypes.data, 0 = NULL): no
        per-call conversion of numpy arrays into ctypes pointers (~10 us of Python per call).  The caller vouches for sizes."""
        f = self.__dict__.get("_ifh_raw")
        if f is None:
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
            f = self._ifh_raw = C.cast(self.L.tf_integrate_frame_host, proto)
        self._ck(f(self.h, depth_addr, rgba_addr or None, pose_addr, pose_inv16_addr or None, int(frame_id)))

    def host_register(self, array):
        """page-lock a caller-owned numpy array in place: host frames passed from it are uploaded without a staging copy"""
        self._ck(self.L.tf_host_register(self.h, C.c_void_p(array.ctypes.data), int(array.nbytes)))

    def host_unregister(self, array):
        self._ck(self.L.tf_host_unregister(self.h, C.c_void_p(array.ctypes.data)))

    def host_frame_times(self, reset=False):
        """host microseconds per call of integrate_frame_host since create / the last reset, by phase"""
        o = (C.c_double * 7)()
        self._ck(self.L.tf_host_frame_times(self.h, o, int(reset)))
        n = o[0] or 1.0
        return {"calls": int(o[0]), "wait_for_device_us": o[1] / n, "wait_for_upload_us": o[2] / n, "staging_copy_us": o[3] / n,
                "upload_enqueue_us": o[4] / n, "launches_us": o[5] / n, "launches_that_waited_for_an_upload": int(o[6])}

    def integrate_frame_host_rgb(self, depth, rgb, color_valid, pose, pose_inv16=None, frame_id=0):
        """the same with Frame::rgb (u8[H][W][3]) and Frame::colorValidFlag (u8[H][W], or None = every pixel valid): the
        caller's RGBA staging loop runs on the device"""
        depth = _f32(depth)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        npix = self.cam.width * self.cam.height
        self._check_images(depth)
        if rgb.size != 3 * npix:
            raise TFError(TF_ERR_INVALID, "rgb has %d bytes, the bound camera needs %d" % (rgb.size, 3 * npix))
        cv = None if color_valid is None else np.ascontiguousarray(color_valid, np.uint8)
        if cv is not None and cv.size != npix:
            raise TFError(TF_ERR_INVALID, "color_valid has %d bytes, the bound camera needs %d" % (cv.size, npix))
        pose = _f32(pose).reshape(12)
        T = None if pose_inv16 is None else _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_integrate_frame_host_rgb(self.h, _p(depth, C.c_float), _p(rgb, C.c_uint8), _p(cv, C.c_uint8),
                                                    _p(pose, C.c_float), _p(T, C.c_float), int(frame_id)))

    def _check_images(self, depth, rgba=None):
        """the C side copies W*H*4 bytes from each pointer: a wrongly sized array is an error here, not an
        out-of-bounds host read there"""
        npix = self.cam.width * self.cam.height
        if depth is not None and depth.size != npix:
            raise TFError(TF_ERR_INVALID, "depth has %d pixels, the bound camera %dx%d" % (depth.size, self.cam.width, self.cam.height))
        if rgba is not None and rgba.size != 4 * npix:
            raise TFError(TF_ERR_INVALID, "rgba has %d bytes, the bound camera needs %d" % (rgba.size, 4 * npix))

    def host_frame_buffers(self):
        """numpy views (depth f32[H,W], rgba u8[H,W,4]) of the pinned slot the next integrate_frame_host uploads from."""
        d, c = C.POINTER(C.c_float)(), C.POINTER(C.c_uint8)()
        self._ck(self.L.tf_host_frame_buffers(self.h, C.byref(d), C.byref(c)))
        H, W = self.cam.height, self.cam.width
        return (np.ctypeslib.as_array(d, shape=(H, W)), np.ctypeslib.as_array(c, shape=(H, W, 4)))

    def sync(self):
        self._ck(self.L.tf_sync(self.h))

    # -- state access
    def has_chunk(self, cid):
        cid = np.ascontiguousarray(cid, np.int32)
        out = C.c_int(0)
        self._ck(self.L.tf_has_chunk(self.h, _p(cid, C.c_int32), C.byref(out)))
        return bool(out.value)

    def get_chunks(self, ids):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        n = len(ids)
        sdf = np.zeros((n, 512), np.float32)
        w = np.zeros((n, 512), np.float32)
        col = np.zeros((n, 2048), np.uint16)
        self._ck(self.L.tf_chunks_download(self.h, _p(ids, C.c_int32), n, _p(sdf, C.c_float),
                                           _p(w, C.c_float), _p(col, C.c_uint16)))
        return sdf, w, col

    def get_chunk(self, cid):
        s, w, c = self.get_chunks(np.asarray(cid, np.int32).reshape(1, 3))
        return s[0], w[0], c[0]

    def set_chunk(self, cid, sdf, w, col):
        cid = np.ascontiguousarray(cid, np.int32)
        col = None if col is None else np.ascontiguousarray(col, np.uint16)
        sdf = None if sdf is None else _f32(sdf)
        w = None if w is None else _f32(w)
        self._ck(self.L.tf_chunk_upload(self.h, _p(cid, C.c_int32), _p(sdf, C.c_float),
                                        _p(w, C.c_float), _p(col, C.c_uint16)))

    def _list(self, fn):
        n = C.c_int64(0)
        self._ck(fn(self.h, None, 0, C.byref(n)))
        ids = np.zeros((max(n.value, 1), 3), np.int32)
        self._ck(fn(self.h, _p(ids, C.c_int32), n.value, C.byref(n)))
        return ids[:n.value]

    def list_chunks(self):
        return self._list(self.L.tf_list_chunks)

    def dirty(self):
        return self._list(self.L.tf_list_dirty)

    def clear_dirty(self):
        self._ck(self.L.tf_clear_dirty(self.h))

    def stats(self):
        st = Stats()
        self._ck(self.L.tf_get_stats(self.h, C.byref(st)))
        return st

    # -- meshing (Chisel::UpdateMeshes / CompressMeshes, ChunkManager::allMeshes)
    def update_meshes(self):
        n = C.c_int64(0)
        self._ck(self.L.tf_update_meshes(self.h, C.byref(n)))
        return n.value

    def list_meshes(self):
        return self._list(self.L.tf_list_meshes)

    def check_summaries(self):
        """-> (alive chunks, chunks whose filter summary lacks a class their voxels hold, chunks with a stale class)"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._ck(self.L.tf_check_summaries(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def check_neighbours(self):
        """-> i64[6]: rows, non-zero words, wrong words (must be 0), trusted rows, trusted "none" words whose chunk exists
        (must be 0), 0 -- the neighbour table against the chunk hash"""
        out = np.zeros(6, np.int64)
        self._ck(self.L.tf_check_neighbours(self.h, _p(out, C.c_int64)))
        return out

    def mesh_counts(self, ids):
        """-> (n_vertices i32[n], n_indices i32[n], adj u8[n,6], simplified u8[n])"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        n = len(ids)
        nv = np.zeros(max(n, 1), np.int32); ni = np.zeros(max(n, 1), np.int32)
        adj = np.zeros((max(n, 1), 6), np.uint8); simp = np.zeros(max(n, 1), np.uint8)
        self._ck(self.L.tf_mesh_counts(self.h, _p(ids, C.c_int32), n, _p(nv, C.c_int32), _p(ni, C.c_int32),
                                       _p(adj, C.c_uint8), _p(simp, C.c_uint8)))
        return nv[:n], ni[:n], adj[:n], simp[:n]

    def get_meshes(self, ids):
        """Mesh::vertices / normals / colors / indices of the listed chunks ->
        (voff i64[n+1], ioff i64[n+1], verts [nv,3], normals, colors, indices u32[ni], adj, simplified)"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        n = len(ids)
        nv, ni, adj, simp = self.mesh_counts(ids)
        voff = np.concatenate([[0], np.cumsum(nv)]).astype(np.int64)
        ioff = np.concatenate([[0], np.cumsum(ni)]).astype(np.int64)
        V = np.zeros((max(int(voff[-1]), 1), 3), np.float32); N = np.zeros_like(V); Cc = np.zeros_like(V)
        I = np.zeros(max(int(ioff[-1]), 1), np.uint32)
        self._ck(self.L.tf_meshes_download(self.h, _p(ids, C.c_int32), n, _p(voff, C.c_int64), _p(ioff, C.c_int64),
                                           _p(V, C.c_float), _p(N, C.c_float), _p(Cc, C.c_float), _p(I, C.c_uint32)))
        return voff, ioff, V[:voff[-1]], N[:voff[-1]], Cc[:voff[-1]], I[:ioff[-1]], adj, simp

    def compress_meshes(self):
        """Chisel::CompressMeshes(meshesToUpdate) -> chunksToUpdate (ascending id); clears the dirty set."""
        n = C.c_int64(0)
        cap = 1 << 16
        while True:
            ids = np.empty((cap, 3), np.int32)
            rc = self.L.tf_compress_meshes(self.h, _p(ids, C.c_int32), cap, C.byref(n))
            if rc == TF_ERR_CAPACITY and n.value > cap:
                raise TFError(rc, "compress_meshes: list larger than %d" % cap)
            self._ck(rc)
            return ids[:n.value].copy()

    # -- measurement
    def profile_enable(self, kinds=PROF_NAMES):
        """kinds: iterable of kernel names from PROF_NAMES (empty / None = off)."""
        mask = 0
        for k in (kinds or ()):
            mask |= 1 << PROF_NAMES.index(k)
        self._ck(self.L.tf_profile_enable(self.h, mask))

    def profile_get(self, reset=True):
        p = Profile()
        self._ck(self.L.tf_profile_get(self.h, C.byref(p), int(reset)))
        return {PROF_NAMES[i]: (p.ms[i], p.launches[i]) for i in range(len(PROF_NAMES))}

    # -- the keyframe unit (MobileFusion::tsdfFusion as one asynchronous call)
    @staticmethod
    def unit_group(kf_id, keyframe, local=(), old_keyframe_pose=None, old_local_poses=()):
        """keyframe = (d_depth, d_rgba, d_quality, pose); local = [(d_depth, pose), ...] (device pointers)"""
        g = UnitGroup()
        g.kf_id = int(kf_id)
        g.n_local = len(local)

        def fill(fr, dd, dc, dq, pose):
            fr.d_depth, fr.d_rgba, fr.d_quality = dd or None, dc or None, dq or None
            fr.pose[:] = list(_f32(pose).reshape(12))
        fill(g.keyframe, *keyframe)
        for i, (dd, pose) in enumerate(local):
            fill(g.local[i], dd, 0, 0, pose)
        if old_keyframe_pose is not None:
            g.old_keyframe_pose[:] = list(_f32(old_keyframe_pose).reshape(12))
        for i, p in enumerate(old_local_poses):
            g.old_local_pose[i][:] = list(_f32(p).reshape(12))
        return g

    def keyframe_unit(self, fresh=None, moved=(), texture=False, pose_inv16=None):
        arr = (UnitGroup * max(1, len(moved)))(*moved)
        T = None if pose_inv16 is None else _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_keyframe_unit_device(self.h, C.byref(fresh) if fresh is not None else None, arr, len(moved),
                                                int(bool(texture)), _p(T, C.c_float)))

    def keyframe_unit_stats(self):
        """{capacity, top, compactions, reuses, regions} of the keyframes' validChunks store"""
        out = (C.c_int64 * 5)()
        self._ck(self.L.tf_keyframe_unit_stats(self.h, out))
        return dict(zip(("capacity", "top", "compactions", "reuses", "regions"), [int(x) for x in out]))

    def keyframe_unit_stats_ex(self):
        """{slots, keyframes, doublings, arena} of the keyframes' validChunks store (grows on demand)"""
        out = (C.c_int64 * 4)()
        self._ck(self.L.tf_keyframe_unit_stats_ex(self.h, out))
        return dict(zip(("slots", "keyframes", "doublings", "arena"), [int(x) for x in out]))

    # -- Chunk::observations on the device and the exports TexMap consumes
    def observations_record(self, keyframe_id):
        self._ck(self.L.tf_observations_record(self.h, int(keyframe_id)))

    def observations_retract(self, keyframe_id, ids):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        self._ck(self.L.tf_observations_retract(self.h, int(keyframe_id), _p(ids, C.c_int32), len(ids)))

    def export_datacost(self, ids, frame_index, frames_to_update=()):
        """table [n, 1 + len(frames_to_update)] of observation qualities (0 = none), TexMap::update_datacost's input"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        fr = np.ascontiguousarray(frames_to_update, np.int32).reshape(-1)
        out = np.zeros((len(ids), 1 + len(fr)), np.float32)
        self._ck(self.L.tf_export_datacost(self.h, _p(ids, C.c_int32), len(ids), int(frame_index),
                                           _p(fr, C.c_int32) if len(fr) else None, len(fr), _p(out, C.c_float)))
        return out

    def export_adjacency(self, ids):
        """edges [m, 4] = (index into ids, neighbour id) of TexMap::update_chunkgraph"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        out = np.zeros((6 * max(1, len(ids)), 4), np.int32)
        n = C.c_int64(0)
        self._ck(self.L.tf_export_adjacency(self.h, _p(ids, C.c_int32), len(ids), _p(out, C.c_int32), len(out), C.byref(n)))
        return out[:n.value]

    def profile_calibrate(self, n_pairs=200):
        """microseconds a HIP-event pair around an empty launch reads (the floor inside every profile_get time)"""
        us = C.c_double(0.0)
        self._ck(self.L.tf_profile_calibrate(self.h, int(n_pairs), C.byref(us)))
        return us.value

    def debug_phase_raw(self):
        out = np.zeros((16384, 16), np.uint64)
        self._ck(self.L.tf_debug_phase_raw(self.h, _p(out, C.c_uint64), out.size))
        return out

    # -- multi-GPU boundary exchange
    def boundary_pack(self, d_buf, cap):
        n = C.c_int64(0)
        self._ck(self.L.tf_boundary_pack(self.h, C.c_void_p(d_buf), cap, C.byref(n)))
        return n.value

    def boundary_pack_async(self, d_buf, cap, d_count):
        """Pack without a host round trip; the record count lands in the device u32 at d_count."""
        self._ck(self.L.tf_boundary_pack_async(self.h, C.c_void_p(d_buf), cap, C.c_void_p(d_count)))

    def boundary_unpack(self, d_buf, n):
        self._ck(self.L.tf_boundary_unpack(self.h, C.c_void_p(d_buf), n))

    def boundary_pack_block(self, d_block, cap):
        """[count | cap records] block of this rank's updated ghost-band chunks; asynchronous."""
        self._ck(self.L.tf_boundary_pack_block(self.h, C.c_void_p(d_block), cap))

    def boundary_unpack_blocks(self, d_blocks, n_blocks, own_block, cap, join_dirty=False):
        self._ck(self.L.tf_boundary_unpack_blocks(self.h, C.c_void_p(d_blocks), n_blocks, own_block, cap, int(join_dirty)))

    def texture_frame_device_phase(self, pose_inv16, frame_id, phase):
        """phase 1: dirty set + interior meshes (before the caller's exchange); phase 2: boundary meshes + pending patch stage"""
        T = _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_texture_frame_device_phase(self.h, _p(T, C.c_float), int(frame_id), int(phase)))

    def comm_exchange_overlap(self, on):
        self._ck(self.L.tf_comm_exchange_overlap(self.h, 1 if on else 0))

    def texture_frame_device(self, pose_inv16, frame_id):
        T = _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_texture_frame_device(self.h, _p(T, C.c_float), int(frame_id)))

    # -- RCCL inside the library
    def comm_init(self, rank, nranks, unique_id):
        """unique_id: 128 bytes from comm_unique_id() of one rank, distributed by the caller."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._ck(self.L.tf_comm_init(self.h, rank, nranks, C.cast(buf, C.c_void_p)))

    def boundary_pack_bands(self, d_block_down, d_block_up, cap):
        """the ghost band as two blocks: what the rank below / the rank above reads"""
        self._ck(self.L.tf_boundary_pack_bands(self.h, C.c_void_p(d_block_down), C.c_void_p(d_block_up), cap))

    def host_frame_set_deferral(self, on):
        """on = False: tf_integrate_frame_host integrates a frame in the call that brings it (no latency, two more
        selection-only launches per frame)"""
        self._ck(self.L.tf_host_frame_set_deferral(self.h, 1 if on else 0))

    def host_frame_set_async(self, on):
        """on: calls out of registered buffers return before their upload is through; host_frame_fence() before a buffer is reused"""
        self._ck(self.L.tf_host_frame_set_async(self.h, 1 if on else 0))

    def host_frame_fence(self):
        self._ck(self.L.tf_host_frame_fence(self.h))

    def host_frame_deferral(self):
        """(frames tf_integrate_frame_host runs behind its caller, staging slots of its ring)"""
        a, b = C.c_int32(0), C.c_int32(0)
        self._ck(self.L.tf_host_frame_deferral(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def boundary_band_bounds(self, cap):
        """record capacities (send_down, send_up, recv_from_below, recv_from_above) of the sized neighbour exchange for
        the frame integrated last: the same numbers on both sides of every transfer, from the frame's own selection"""
        out = (C.c_int64 * 4)()
        self._ck(self.L.tf_boundary_band_bounds(self.h, cap, out))
        return tuple(int(x) for x in out)

    def boundary_pack_bands2(self, d_block_down, cap_down, d_block_up, cap_up):
        self._ck(self.L.tf_boundary_pack_bands2(self.h, C.c_void_p(d_block_down), cap_down, C.c_void_p(d_block_up), cap_up))

    def boundary_unpack_pair(self, d_from_below, cap_below, d_from_above, cap_above, join_dirty=True):
        self._ck(self.L.tf_boundary_unpack_pair(self.h, C.c_void_p(d_from_below), cap_below, C.c_void_p(d_from_above),
                                                cap_above, 1 if join_dirty else 0))

    def comm_stats_ex(self):
        out = (C.c_int64 * 8)()
        self._ck(self.L.tf_comm_stats_ex(self.h, out))
        d = dict(zip(("exchanges", "bytes_sent", "bytes_received", "records_sent", "records_received", "bound_records",
                      "mode", "checked"), (int(x) for x in out)))
        d["overlapped"] = d["checked"] >> 1  # exchanges that ran on the second stream next to an interior mesh pass
        d["checked"] &= 1
        return d

    def comm_exchange_mode(self, mode):
        """0 = neighbour send / receive pairs (default), 1 = all-gather"""
        self._ck(self.L.tf_comm_exchange_mode(self.h, int(mode)))

    def comm_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(self.L.tf_comm_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def exchange_boundary(self, cap):
        self._ck(self.L.tf_exchange_boundary(self.h, cap))

    def comm_exchange_every_frame(self, cap):
        self._ck(self.L.tf_comm_exchange_every_frame(self.h, cap))

    def integrate_depth_group(self, d_depth_ptrs, poses12, ids, needs, flag=1):
        """n depth-only frames (device pointers) over the list `ids` in one visit per chunk; needs is updated in place."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        P = _f32(poses12).reshape(-1, 12)
        nf = len(d_depth_ptrs)
        assert P.shape[0] == nf and needs.dtype == np.uint8 and len(needs) == len(ids)
        arr = (C.c_void_p * nf)(*[int(p) for p in d_depth_ptrs])
        self._ck(self.L.tf_integrate_depth_group(self.h, nf, arr, _p(P, C.c_float), _p(ids, C.c_int32), len(ids), int(flag),
                                                 _p(needs, C.c_uint8)))

    def integrate_depth_group_host(self, depths, poses12, ids, needs, flag=1):
        """the same with host depth images"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        P = _f32(poses12).reshape(-1, 12)
        keep = [_f32(d) for d in depths]
        for d in keep:
            self._check_images(d)
        arr = (C.c_void_p * len(keep))(*[d.ctypes.data for d in keep])
        self._ck(self.L.tf_integrate_depth_group_host(self.h, len(keep), arr, _p(P, C.c_float), _p(ids, C.c_int32), len(ids),
                                                      int(flag), _p(needs, C.c_uint8)))

    # -- frame pre-processing on device-resident images (raw device pointers; BasicAPI.cpp:378-905)
    def pre_normal_map(self, d_depth, d_normal):
        self._ck(self.L.tf_pre_normal_map(self.h, d_depth, d_normal))

    def pre_refine_depth_normal(self, d_normal, d_depth):
        self._ck(self.L.tf_pre_refine_depth_normal(self.h, d_normal, d_depth))

    def pre_color_valid(self, d_normal, d_flag):
        self._ck(self.L.tf_pre_color_valid(self.h, d_normal, d_flag))

    def pre_color_quality(self, d_depth, d_normal, d_rgb, d_quality):
        self._ck(self.L.tf_pre_color_quality(self.h, d_depth, d_normal, d_rgb, d_quality))

    def pre_frame_depth(self, d_depth_u16, d_refined, maximum_depth, depth_scale, d=9, sigma_color=0.03, sigma_space=10.0):
        """DatasetWrapper::framePreprocess on a device u16 depth map (updated in place) -> d_refined f32 (may be 0)"""
        self._ck(self.L.tf_pre_frame_depth(self.h, d_depth_u16, d_refined, maximum_depth, depth_scale, d, sigma_color,
                                           sigma_space))

    def pre_refine_newframe(self, d_depth_ref, d_depth_new, T12):
        T = _f32(T12).reshape(12)
        self._ck(self.L.tf_pre_refine_newframe(self.h, d_depth_ref, d_depth_new, _p(T, C.c_float)))

    def pre_refine_keyframe(self, d_depth_ref, d_weight_ref, d_depth_new, T12):
        T = _f32(T12).reshape(12)
        rounds = C.c_int32(0)
        self._ck(self.L.tf_pre_refine_keyframe(self.h, d_depth_ref, d_weight_ref, d_depth_new, _p(T, C.c_float),
                                               C.byref(rounds)))
        return int(rounds.value)

    # -- atlas (device-resident meshes and patches)
    def keyframe_cache(self, kf_id, rgb, depth, pose_inv16=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = _f32(depth)
        self._ck(self.L.tf_keyframe_cache(self.h, kf_id, _p(rgb, C.c_uint8), _p(depth, C.c_float)))
        if pose_inv16 is not None:
            self.keyframe_set_pose(kf_id, pose_inv16)

    def keyframe_cache_device(self, kf_id, d_rgb, d_depth, stride=3, pose_inv16=None):
        self._ck(self.L.tf_keyframe_cache_device(self.h, kf_id, C.c_void_p(d_rgb), stride, C.c_void_p(d_depth)))
        if pose_inv16 is not None:
            self.keyframe_set_pose(kf_id, pose_inv16)

    def keyframe_set_pose(self, kf_id, pose_inv16):
        T = _f32(pose_inv16).reshape(16)
        self._ck(self.L.tf_keyframe_set_pose(self.h, kf_id, _p(T, C.c_float)))

    def keyframe_release(self, kf_id):
        self._ck(self.L.tf_keyframe_release(self.h, kf_id))

    def atlas_patch_size(self):
        a, b = C.c_int32(0), C.c_int32(0)
        self._ck(self.L.tf_atlas_patch_size(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def atlas_loc_next(self):
        t = C.c_uint64(0)
        self._ck(self.L.tf_atlas_loc_next(self.h, C.byref(t)))
        return t.value

    def meshes_upload(self, ids, voff, ioff, verts, normals, colors, indices):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        voff = np.ascontiguousarray(voff, np.int64); ioff = np.ascontiguousarray(ioff, np.int64)
        indices = np.ascontiguousarray(indices, np.uint32)
        self._ck(self.L.tf_meshes_upload(self.h, _p(ids, C.c_int32), len(ids), _p(voff, C.c_int64), _p(ioff, C.c_int64),
                                         _p(_f32(verts), C.c_float), _p(_f32(normals), C.c_float),
                                         _p(_f32(colors), C.c_float), _p(indices, C.c_uint32)))

    def generate_patches(self, ids, labels):
        """Chisel::GeneratePatches -> (rc, (hot_start, hot_end)); rc == TF_ERR_ATLAS_FULL is GeneratePatches' -1."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        labels = np.ascontiguousarray(labels, np.int32)
        hot = np.zeros(2, np.uint64)
        rc = self.L.tf_generate_patches(self.h, _p(ids, C.c_int32), len(ids), _p(labels, C.c_int32), _p(hot, C.c_uint64))
        if rc != TF_ERR_ATLAS_FULL:
            self._ck(rc)
        return rc, (int(hot[0]), int(hot[1]))

    def compensate_color(self):
        n = C.c_int64(0)
        self._ck(self.L.tf_compensate_color(self.h, C.byref(n)))
        return n.value

    def update_atlas(self, ids):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        self._ck(self.L.tf_update_atlas(self.h, _p(ids, C.c_int32), len(ids)))

    def draw_meshes(self):
        """Chisel::DrawMeshes -> (vertices f32[n,12], indices u32[m])"""
        nv, ni = C.c_int64(0), C.c_int64(0)
        dummy_v = np.zeros(12, np.float32); dummy_i = np.zeros(1, np.uint32)
        rc = self.L.tf_draw_meshes(self.h, _p(dummy_v, C.c_float), _p(dummy_i, C.c_uint32), 0, 0, C.byref(nv), C.byref(ni))
        if rc not in (TF_OK, TF_ERR_CAPACITY):
            self._ck(rc)
        V = np.zeros((max(nv.value, 1), 12), np.float32); I = np.zeros(max(ni.value, 1), np.uint32)
        self._ck(self.L.tf_draw_meshes(self.h, _p(V, C.c_float), _p(I, C.c_uint32), nv.value, ni.value,
                                       C.byref(nv), C.byref(ni)))
        return V[:nv.value], I[:ni.value]

    def get_patches(self, ids):
        """Patch mirrors of the listed chunks -> dict(voff, texloc, frameid, bbox, flags, ratio, texcoord, texcolor, labs)"""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, 3)
        n = len(ids)
        nv, _, _, _ = self.mesh_counts(ids)
        voff = np.concatenate([[0], np.cumsum(nv)]).astype(np.int64)
        tot = int(voff[-1])
        texloc = np.zeros(max(n, 1), np.uint64); frameid = np.zeros(max(n, 1), np.int32)
        bbox = np.zeros((max(n, 1), 4), np.int32); flags = np.zeros(max(n, 1), np.int32)
        ratio = np.zeros((max(n, 1), 2), np.float32)
        tc = np.zeros((max(tot, 1), 2), np.float32); tcol = np.zeros((max(tot, 1), 3), np.float32)
        labs = np.zeros((max(tot, 1), 3), np.float32)
        self._ck(self.L.tf_patches_download(self.h, _p(ids, C.c_int32), n, _p(voff, C.c_int64), _p(texloc, C.c_uint64),
                                            _p(frameid, C.c_int32), _p(bbox, C.c_int32), _p(flags, C.c_int32),
                                            _p(ratio, C.c_float), _p(tc, C.c_float), _p(tcol, C.c_float),
                                            _p(labs, C.c_float)))
        return dict(voff=voff, texloc=texloc[:n], frameid=frameid[:n], bbox=bbox[:n], flags=flags[:n], ratio=ratio[:n],
                    texcoord=tc[:tot], texcolor=tcol[:tot], labs=labs[:tot])

    def texture_stats(self):
        st = TextureStats()
        self._ck(self.L.tf_get_texture_stats(self.h, C.byref(st)))
        return st

    def atlas_rows(self, row0, row1, width):
        out = np.zeros((row1 - row0, width, 3), np.uint8)
        self._ck(self.L.tf_atlas_download_rows(self.h, row0, row1, _p(out, C.c_uint8)))
        return out

    def atlas_snapshot_rows(self, row0, row1, width):
        """-> (rows, write_seq, frame_id): tf_atlas_snapshot_rows -- callable from a thread other than the one that drives
        the handle (ctypes releases the GIL for the call)"""
        out = np.zeros((row1 - row0, width, 3), np.uint8)
        seq, fid = C.c_int64(0), C.c_int32(-1)
        self._ck(self.L.tf_atlas_snapshot_rows(self.h, row0, row1, _p(out, C.c_uint8), C.byref(seq), C.byref(fid)))
        return out, seq.value, fid.value
