"""Shared helpers of the parity tests: scene set-up on both sides (oracle = checker, HIP = product)."""
import numpy as np

from oracle import api as O
from texturefusion_amd import capi, synth

RES5 = np.float32(0.005)
RES10 = np.float32(0.01)


def make_pair(res=RES5, cam=None, **kw):
    cam = cam or synth.Camera()
    ig = O.default_integrator()
    ov = O.Volume(res, O.camera_from(cam), ig)
    gv = capi.Volume(res, cam, **kw)
    return ov, gv, cam, ig


def assert_chunks_equal(ov, gv, ids, what=""):
    ids = np.asarray(ids, np.int32).reshape(-1, 3)
    if len(ids) == 0:
        return
    for b0 in range(0, len(ids), 8192):  # (in blocks: a hall has 10^5 chunks of 8 KiB)
        blk = ids[b0:b0 + 8192]
        gs, gw, gc = gv.get_chunks(blk)
        for i, cid in enumerate(blk):
            os_, ow, oc = ov.get_chunk(cid)
            # bit-exact: compare raw bits so that -0.0 / NaN payloads would be caught too
            assert np.array_equal(os_.view(np.uint32), gs[i].view(np.uint32)), "%s sdf differs in chunk %s" % (what, cid)
            assert np.array_equal(ow.view(np.uint32), gw[i].view(np.uint32)), "%s weight differs in chunk %s" % (what, cid)
            assert np.array_equal(oc, gc[i]), "%s colour differs in chunk %s" % (what, cid)


def sorted_ids(ids):
    ids = np.asarray(ids, np.int32).reshape(-1, 3)
    if len(ids) == 0:
        return ids
    order = np.lexsort((ids[:, 2], ids[:, 1], ids[:, 0]))
    return ids[order]


class HipBuffer:
    """Raw device buffer through the HIP runtime the product library already loaded (no torch)."""

    def __init__(self, nbytes):
        import ctypes as C
        capi.lib()
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self._hip.hipFree.argtypes = [C.c_void_p]
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        p = C.c_void_p()
        rc = self._hip.hipMalloc(C.byref(p), nbytes)
        assert rc == 0, "hipMalloc failed: %d" % rc
        self.ptr = p.value
        self.nbytes = nbytes

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        rc = self._hip.hipMemcpy(self.ptr, arr.ctypes.data, arr.nbytes, 1)  # hipMemcpyHostToDevice
        assert rc == 0
        return self

    def to_host(self, nbytes=None):
        n = self.nbytes if nbytes is None else nbytes
        out = np.empty(n, np.uint8)
        rc = self._hip.hipMemcpy(out.ctypes.data, self.ptr, n, 2)  # hipMemcpyDeviceToHost
        assert rc == 0
        return out

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None
