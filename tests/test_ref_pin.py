"""oracle/_ref -- the part of the reference that compiles here, unmodified and without stand-ins
(oracle/ref_shim.cpp: QuadraticTruncator / ConstantWeighter, chisel::parallel_for, SparseMat) -- against the oracle's
restatements and the host mirror's classes.  This ties SURVEY.md s.8 row a8, the thread policy of rows a6 / f-1 and
the data-cost container of row f-4 to compiled reference code.  It pins NOTHING else: K-A, the selection, the mesher
and the atlas need Eigen / OpenCV / Sophus and stay unpinned.

The library is built from /root/reference when that exists (this container); elsewhere the prebuilt
oracle/_ref/libtf_ref.so is used, and the tests skip when neither is there."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import api as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libtf_ref.so")
MIR_SO = os.path.join(ROOT, "tests", "cpp", "libmirror_shim.so")

fp = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)


def _bind_sm(L, pre):
    g = lambda n: getattr(L, pre + n)
    g("sm_new").restype = C.c_void_p
    g("sm_free").argtypes = [C.c_void_p]
    for n in ("sm_cols", "sm_rows", "sm_nnz"):
        g(n).restype = C.c_uint64
        g(n).argtypes = [C.c_void_p]
    g("sm_add_value").argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_float]
    g("sm_set_value").argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_float]
    g("sm_resize").argtypes = [C.c_void_p, C.c_uint64]
    g("sm_clear").argtypes = [C.c_void_p]
    g("sm_remove_node").argtypes = [C.c_void_p, C.c_uint64]
    g("sm_remove_observation").argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    g("sm_col").restype = C.c_uint64
    g("sm_col").argtypes = [C.c_void_p, C.c_uint64, u64p, fp, C.c_uint64]
    g("truncation_n").argtypes = [C.c_float] * 4 + [fp, fp, C.c_int64]
    g("weight_n").argtypes = [C.c_float, fp, fp, C.c_int64]


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libtf_ref.so not built and /root/reference absent")
    L = C.CDLL(REF_SO)
    _bind_sm(L, "tfref_")
    L.tfref_truncation.restype = C.c_float
    L.tfref_truncation.argtypes = [C.c_float] * 5
    L.tfref_parallel_for_groups.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    return L


@pytest.fixture(scope="module")
def mirror():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "libmirror_shim.so"], check=True, capture_output=True)
    L = C.CDLL(MIR_SO)
    _bind_sm(L, "tfmir_")
    return L


def _trunc_inputs(n, seed):
    rng = np.random.default_rng(seed)
    z = np.concatenate([
        rng.uniform(-6.0, 6.0, n).astype(np.float32),           # the depth range of the path (and behind the camera)
        np.float32(10.0) ** rng.uniform(-30, 30, n // 4).astype(np.float32),
        rng.integers(0, 2 ** 32, n // 4, dtype=np.uint64).astype(np.uint32).view(np.float32),  # any bit pattern
        np.array([0.0, -0.0, 1.5, 3.0, 0.4, 5.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38], np.float32)])
    return np.ascontiguousarray(z)


def _call_n(fn, params, x):
    out = np.empty_like(x)
    fn(*[C.c_float(p) for p in params], x.ctypes.data_as(fp), out.ctypes.data_as(fp), len(x))
    return out


PARAM_SETS = [(0.0019, 0.00152, 0.001504, 6.0),      # MobileFusion.h:245-247
              (0.0019, 0.00152, 0.001504, 1.0), (0.0, 0.0, 0.03, 1.0), (-0.01, 0.02, -0.005, 8.0),
              (1e-3, -4e-3, 1e-3, 3.0)]


def test_truncation_equals_compiled_reference(ref, mirror):
    """tfo_truncation and the mirror's QuadraticTruncator vs chisel::QuadraticTruncator::GetTruncationDistance called
    through the base-class pointer (ProjectionIntegrator.cpp:91), bit for bit, 3e4 inputs per parameter set."""
    L = O.lib()
    z = _trunc_inputs(20000, 1)
    for ps in PARAM_SETS:
        ps32 = [np.float32(p) for p in ps]
        want = _call_n(ref.tfref_truncation_n, ps32, z)
        ig = O.Integrator(*ps32, np.float32(1.0))
        got = np.array([L.tfo_truncation(C.byref(ig), C.c_float(v)) for v in z], np.float32)
        nan = np.isnan(want)
        assert np.array_equal(nan, np.isnan(got))
        assert np.array_equal(want[~nan].view(np.uint32), got[~nan].view(np.uint32)), ps
        mir = _call_n(mirror.tfmir_truncation_n, ps32, z)
        assert np.array_equal(want[~nan].view(np.uint32), mir[~nan].view(np.uint32)), ps
        assert np.array_equal(nan, np.isnan(mir))
    assert ref.tfref_truncation(*[C.c_float(np.float32(p)) for p in PARAM_SETS[0]], C.c_float(1.5)) > 0


def test_weight_equals_compiled_reference(ref, mirror):
    """tfo_weight / the mirror's ConstantWeighter vs chisel::ConstantWeighter::GetWeight (ProjectionIntegrator.cpp:92)."""
    L = O.lib()
    L.tfo_weight.restype = C.c_float
    L.tfo_weight.argtypes = [C.POINTER(O.Integrator), C.c_float]
    tr = np.abs(_trunc_inputs(20000, 2))
    for w in (1.0, 0.5, 3.0, 1e-3):
        w32 = np.float32(w)
        want = _call_n(ref.tfref_weight_n, [w32], tr)
        ig = O.Integrator(*[np.float32(p) for p in PARAM_SETS[0]], w32)
        got = np.array([L.tfo_weight(C.byref(ig), C.c_float(v)) for v in tr], np.float32)
        mir = _call_n(mirror.tfmir_weight_n, [w32], tr)
        nan = np.isnan(want)
        for other in (got, mir):
            assert np.array_equal(nan, np.isnan(other))
            assert np.array_equal(want[~nan].view(np.uint32), other[~nan].view(np.uint32)), w


def test_chunk_scalars_use_the_pinned_pieces(ref):
    """The per-chunk truncation / weight K-A works with (tfo_chunk_scalars, ProjectionIntegrator.cpp:88-92) are the
    reference functions applied to the oracle's originInCamera.z -- the z itself depends on Eigen's dot order (unpinned)."""
    ig = O.default_integrator()
    rng = np.random.default_rng(5)
    for _ in range(200):
        pose = np.eye(4, dtype=np.float32)[:3].copy()
        pose[:, 3] = rng.uniform(-1, 1, 3).astype(np.float32)
        cid = rng.integers(-40, 40, 3).astype(np.int32)
        oc, tr, w = O.chunk_scalars(ig, pose, cid, np.float32(0.005))
        want_tr = ref.tfref_truncation(*[C.c_float(x) for x in (ig.quad, ig.lin, ig.cons, ig.scale)], C.c_float(oc[2]))
        assert np.float32(want_tr).view(np.uint32) == np.float32(tr).view(np.uint32)
        wv = _call_n(ref.tfref_weight_n, [np.float32(ig.weight)], np.array([tr], np.float32))
        assert wv.view(np.uint32)[0] == np.float32(w).view(np.uint32)


def test_parallel_for_plan_equals_compiled_reference(ref):
    """tfo_parallel_for_plan (the thread cut the oracle's chunk and mesh loops use for the CPU baseline) vs
    chisel::parallel_for itself, run with an index vector as Chisel.h:234 does: number of threads that ran items and
    the exact stretch every item fell into.  (n = 0 is not asked: the reference's loop bound `last - group` on an empty
    vector is a wrapped pointer -- it spawns threads until the process dies -- and Chisel.h:228 returns before it.)"""
    L = O.lib()
    L.tfo_parallel_for_plan.restype = C.c_int
    L.tfo_parallel_for_plan.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    sizes = [1, 2, 999, 1000, 1001, 1999, 2000, 2001, 4606, 4999, 5000, 5001, 11000, 31449, 78000, 100000, 100003]
    total = 0
    for n in sizes:
        worker = np.zeros(max(n, 1), np.int32)
        nt = C.c_int(0)
        used = ref.tfref_parallel_for_groups(n, worker.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nt))
        if nt.value < 1:
            pytest.skip("hardware_concurrency() - 2 < 1 on this host: the reference itself divides by zero")
        group = C.c_int64(0)
        T = L.tfo_parallel_for_plan(n, nt.value, 1000, C.byref(group))
        g = group.value
        assert g == max(1000, n // nt.value)
        assert used == T, (n, nt.value, used, T)
        # item i ran in stretch i // group: the same thread for a whole stretch, another one for the next
        want = np.arange(n) // g
        first = {}
        for i in range(0, n, max(1, g // 3)):
            first.setdefault(int(worker[i]), int(want[i]))
            assert first[int(worker[i])] == int(want[i])
        assert len(set(worker[:n].tolist())) == T
        total += n
    assert total >= 10 ** 4


def _dump(L, pre, h):
    g = lambda n: getattr(L, pre + n)
    cols = g("sm_cols")(h)
    out = [cols, g("sm_rows")(h), g("sm_nnz")(h)]
    rows = np.empty(4096, np.uint64)
    vals = np.empty(4096, np.float32)
    for c in range(cols):
        k = g("sm_col")(h, c, rows.ctypes.data_as(u64p), vals.ctypes.data_as(fp), 4096)
        out.append((c, rows[:k].tolist(), vals[:k].view(np.uint32).tolist()))
    return out


def test_sparse_mat_mirror_equals_compiled_reference(ref, mirror):
    """The mirror's SparseMat (the DataCosts of TexMap, row f-4) vs Structure/sparse_matrix.cpp under one random
    stream of 2e4 operations: every return value, cols / rows / nnz, and every column's ordered content."""
    rng = np.random.default_rng(11)
    a = ref.tfref_sm_new()
    b = mirror.tfmir_sm_new()
    n_ops = 20000
    for k in range(n_ops):
        op = rng.integers(0, 100)
        c = int(rng.integers(0, 60))
        r = int(rng.integers(0, 40))
        v = np.float32(rng.uniform(-5, 5))
        if op < 45:
            assert ref.tfref_sm_add_value(a, c, r, v) == mirror.tfmir_sm_add_value(b, c, r, v)
        elif op < 70:
            ref.tfref_sm_set_value(a, c, r, v); mirror.tfmir_sm_set_value(b, c, r, v)
        elif op < 85:
            ref.tfref_sm_remove_observation(a, c + 5, r); mirror.tfmir_sm_remove_observation(b, c + 5, r)
        elif op < 93:
            ref.tfref_sm_remove_node(a, c + 5); mirror.tfmir_sm_remove_node(b, c + 5)
        elif op < 98:
            n = max(int(ref.tfref_sm_cols(a)), c)      # TexMap only ever grows the matrix (TexMap.cpp:66)
            ref.tfref_sm_resize(a, n); mirror.tfmir_sm_resize(b, n)
        elif op == 98 and k % 7 == 0:
            ref.tfref_sm_clear(a); mirror.tfmir_sm_clear(b)
        if k % 500 == 0 or k == n_ops - 1:
            assert _dump(ref, "tfref_", a) == _dump(mirror, "tfmir_", b), k
    ref.tfref_sm_free(a)
    mirror.tfmir_sm_free(b)


def test_mc_tables_equal_the_reference_source():
    """The nibble-packed marching-cubes triangle table of the oracle and of the product (tools/gen_mc_table.py)
    against the table where the reference keeps it (marching_cubes/MarchingCubes.cpp:28-), and the edge -> corner
    pairs (:288-).  Reads the reference's source text, so it runs in the build container only."""
    import re
    src = "/root/reference/3rd_party/open_chisel/marching_cubes/MarchingCubes.cpp"
    if not os.path.exists(src):
        pytest.skip("/root/reference absent")
    txt = open(src).read()
    body = txt[txt.index("triangleTable[256][16]"):]
    rows = re.findall(r"\{([^{}]*)\}", body[body.index("{") + 1:])[:256]
    want = []
    for r in rows:
        v = [int(x) for x in r.split(",") if x.strip()]
        assert len(v) == 16
        want.append(sum(((0xF if e < 0 else e) << (4 * j)) for j, e in enumerate(v)))
    for path in ("oracle/mc_table.inc", "texturefusion_amd/csrc/tf_mc_table.h"):
        got = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", open(os.path.join(ROOT, path)).read())]
        assert got == want, path
    pairs = txt[txt.index("edgeIndexPairs[12][2]"):]
    pairs = [tuple(int(x) for x in p.split(",")) for p in re.findall(r"\{(\s*\d+\s*,\s*\d+\s*)\}", pairs)[:12]]
    assert pairs == [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    cm = open("/root/reference/Structure/ChunkManager.cpp").read()      # the copy the mesher uses (:68-75)
    cmp_ = cm[cm.index("tempEdgeIndexPairs[12][2]"):]
    assert [tuple(int(x) for x in p.split(",")) for p in re.findall(r"\{(\s*\d+\s*,\s*\d+\s*)\}", cmp_)[:12]] == pairs
    offs = re.search(r"cubeIndexOffsets\s*<<([^;]*);", cm).group(1)
    offs = np.array([int(x) for x in offs.split(",")]).reshape(3, 8).T.tolist()   # column k = corner k (:65-66)
    oc = open(os.path.join(ROOT, "oracle", "tf_oracle.c")).read()
    mk = re.search(r"kCorner\[8\]\[3\]\s*=\s*\{(.*?)\};", oc, re.S)
    assert [[int(x) for x in p.split(",")] for p in re.findall(r"\{([^{}]*)\}", mk.group(1))] == offs
    m = re.search(r"kEdgePair\[12\]\[2\]\s*=\s*\{(.*?)\};", oc, re.S)
    assert m
    if m:
        assert [tuple(int(x) for x in p.split(",")) for p in re.findall(r"\{(\s*\d+\s*,\s*\d+\s*)\}", m.group(1))] == pairs


def test_init_chisel_map_compiles_against_the_mirror(tmp_path):
    """MobileFusion::initChiselMap (GCFusion/MobileFusion.h:205-258) -- the caller's set-up of the path: constructor, the
    integrator's and the camera model's setters -- is cut out of /root/reference AT TEST TIME, wrapped into a struct with
    the members it touches, compiled against texturefusion_amd/host/tf_chisel.hpp and linked against the C ABI (not run:
    the constructor would ask for a device).  tests/cpp/host_mirror_parity.cpp drives the same configuration on the GPU
    with a call sequence of its own.  Nothing of the reference's text is kept in the repository; build container only."""
    hdr = "/root/reference/GCFusion/MobileFusion.h"
    if not os.path.exists(hdr):
        pytest.skip("/root/reference absent")
    lines = open(hdr).read().splitlines()
    first = next(i for i, l in enumerate(lines) if "void initChiselMap(" in l)
    depth, last = 0, None
    for i in range(first, len(lines)):  # the function's text = up to the brace that closes its body
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0 and "{" in "".join(lines[first:i + 1]):
            last = i
            break
    assert last is not None and 40 <= last - first <= 70
    body = "\n".join(lines[first:last + 1])
    src = tmp_path / "init_chisel_map.cpp"
    src.write_text("""
#include <iostream>
#include <cstring>
#include "%s/texturefusion_amd/host/tf_chisel.hpp"
// what the function's text needs from its surroundings and this image lacks: Eigen's Vector3i (the mirror's constructor
// takes any vector indexed with (i)) and the calibration record of GCSLAM/MultiViewGeometry.h
namespace Eigen { typedef chisel::ChunkID Vector3i; }
namespace MultiViewGeometry { struct CameraPara { float c_fx, c_fy, c_cx, c_cy; int width, height; }; }
struct MobileFusion {  // the members the function touches (GCFusion/MobileFusion.h:62-76)
  chisel::ChiselPtr chiselMap;
  chisel::ProjectionIntegrator projectionIntegrator;
  chisel::PinholeCamera cameraModel;
  int chunkSizeX, chunkSizeY, chunkSizeZ;
  float voxelResolution;
  bool useColor;
%s
};
int main() {
  MobileFusion m;
  const MultiViewGeometry::CameraPara camera = {525.0f, 525.0f, 319.5f, 239.5f, 640, 480};
  (void)camera; (void)m;   // (compiled and linked; running it would create a device volume)
  return sizeof(&MobileFusion::initChiselMap) ? 0 : 1;
}
""" % (ROOT, body))
    exe = tmp_path / "init_chisel_map"
    r = subprocess.run(["g++", "-std=c++14", "-O0", "-Wall", str(src), "-o", str(exe), "-L" + os.path.join(ROOT, "texturefusion_amd"),
                        "-ltexfusion_hip", "-Wl,-rpath," + os.path.join(ROOT, "texturefusion_amd"), "-Wl,-rpath,/opt/rocm/lib",
                        "-L/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_gui_texture_upload_compiles_against_the_mirror(tmp_path):
    """MobileFusion::MobileShow's texture upload (GCFusion/MobileFusion.h:404-421) -- the GUI thread hands the hot rows of
    `chiselMap->atlas.texture_buffer` to OpenGL -- is cut out of /root/reference AT TEST TIME (the block of the `if` on
    atlas.hot_end / hot_start), put into a function next to no-op stand-ins for the six GL entry points it calls, compiled
    against texturefusion_amd/host/tf_chisel.hpp and linked against the C ABI: `atlas.hot_start`, `atlas.hot_end`,
    `atlas.texture_buffer.data` and MAX_PATCH_WIDTH are what the mirror has to offer for it to compile unchanged.  Nothing
    of the reference's text is kept in the repository; build container only."""
    hdr = "/root/reference/GCFusion/MobileFusion.h"
    if not os.path.exists(hdr):
        pytest.skip("/root/reference absent")
    lines = open(hdr).read().splitlines()
    first = next(i for i, l in enumerate(lines) if "chiselMap->atlas.hot_end > chiselMap->atlas.hot_start" in l)
    depth, last = 0, None
    for i in range(first, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0 and "{" in "".join(lines[first:i + 1]):
            last = i
            break
    assert last is not None and 12 <= last - first <= 30
    body = "\n".join(lines[first:last + 1])
    assert "texture_buffer" in body and ".data[" in body
    src = tmp_path / "gui_upload.cpp"
    src.write_text("""
#include "%s/texturefusion_amd/host/tf_chisel.hpp"
// stand-ins for what the block calls of OpenGL (this image has no GL): enough for the block to compile as it is
typedef unsigned int GLuint; typedef int GLint; typedef int GLsizei; typedef unsigned int GLenum; typedef long GLsizeiptr;
enum { GL_PIXEL_UNPACK_BUFFER = 1, GL_STREAM_COPY_ARB, GL_TEXTURE_2D, GL_RGB, GL_UNSIGNED_BYTE };
static const void* g_last_upload = nullptr; static long g_last_bytes = 0;
inline void glBindBufferARB(GLenum, GLuint) {}
inline void glBufferDataARB(GLenum, GLsizeiptr bytes, const void* p, GLenum) { g_last_upload = p; g_last_bytes = (long)bytes; }
inline void glBindTexture(GLenum, GLuint) {}
inline void glTexSubImage2D(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void*) {}
struct MobileFusion {  // the members the block touches (GCFusion/MobileFusion.h:62-90)
  chisel::ChiselPtr chiselMap;
  GLuint pbo = 0, texture_model = 0;
  void upload() {
%s
  }
};
int main() {
  MobileFusion m;
  (void)m; (void)g_last_upload; (void)g_last_bytes;  // (compiled and linked; running it would need a device volume)
  return sizeof(&MobileFusion::upload) ? 0 : 1;
}
""" % (ROOT, body))
    exe = tmp_path / "gui_upload"
    r = subprocess.run(["g++", "-std=c++14", "-O0", "-Wall", str(src), "-o", str(exe), "-L" + os.path.join(ROOT, "texturefusion_amd"),
                        "-ltexfusion_hip", "-Wl,-rpath," + os.path.join(ROOT, "texturefusion_amd"), "-Wl,-rpath,/opt/rocm/lib",
                        "-L/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
