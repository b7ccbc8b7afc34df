"""Chisel::CompensateColor (Structure/Chisel.cpp:198-286): analytic known-answer tests of the oracle
restatement (CPU); the HIP path is compared against it in tests/test_gpu_atlas.py.

Tolerance: the eigen-solver of the reference (Eigen's iterative SelfAdjointEigenSolver) is third-party
arithmetic that is not restated ("parity unpinned", 1-ulp class); oracle and product
both use a double-precision Jacobi iteration, reductions differ in association only (sequential vs fixed
tree), so adjusted colours in [0, 1] agree to 2e-5 absolute; clustering, flags and counts are exact."""
import numpy as np
import pytest

from oracle import api as O

TOL = 2e-5


def _batch(seed, n_patches=40, frames=(3, 7, 3, 9), wrong_every=6, adjusted_every=0):
    rng = np.random.default_rng(seed)
    nv = rng.integers(3, 120, n_patches)
    voff = np.concatenate([[0], np.cumsum(nv)]).astype(np.int64)
    fid = np.array([frames[i % len(frames)] for i in range(n_patches)], np.int32)
    wrong = np.array([(i % wrong_every) == wrong_every - 1 for i in range(n_patches)], np.uint8)
    adj = np.zeros(n_patches, np.uint8)
    if adjusted_every:
        adj[::adjusted_every] = 1
    mesh = rng.random((voff[-1], 3)).astype(np.float32) * 0.6 + 0.2
    # the keyframe sees the mesh colours through a per-frame gain / offset (what compensation undoes)
    tex = np.empty_like(mesh)
    for p in range(n_patches):
        g = 0.7 + 0.1 * (fid[p] % 4)
        tex[voff[p]:voff[p + 1]] = np.clip(mesh[voff[p]:voff[p + 1]] * g + 0.05 * (fid[p] % 3), 0, 1)
    return fid, wrong, adj, voff, tex, mesh


def test_transfer_matrix_closed_form_isotropic():
    """cov_src = s^2 I, cov_tar = t^2 I  =>  T = s t / (s + 0.01)^2 I (Chisel.cpp:247-266 by hand)."""
    s, t = 0.2, 0.3
    T = O.color_transfer(np.eye(3) * s * s, np.eye(3) * t * t)
    assert np.allclose(T, np.eye(3) * (s * t / (s + 0.01) ** 2), atol=1e-6)


def test_transfer_matrix_matches_numpy_eigh():
    rng = np.random.default_rng(5)
    for _ in range(20):
        A, B = rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
        cs, ct = (A @ A.T * 0.05).astype(np.float32), (B @ B.T * 0.05).astype(np.float32)
        w, U = np.linalg.eigh(cs.astype(np.float64))
        D = np.diag(np.sqrt(np.maximum(w, 0)))
        wm, Um = np.linalg.eigh(D @ U.T @ ct.astype(np.float64) @ U @ D)
        Di = np.diag(1.0 / (np.sqrt(np.maximum(w, 0)) + 1e-2))
        ref = U @ Di @ Um @ np.diag(np.sqrt(np.maximum(wm, 0))) @ Um.T @ Di @ U.T
        assert np.abs(O.color_transfer(cs, ct) - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())


def test_oracle_clusters_flags_and_statistics():
    fid, wrong, adj, voff, tex, mesh = _batch(1, adjusted_every=5)
    labs, adj2, T, cl = O.color_compensate(fid, wrong, adj, voff, tex, mesh)
    # clusters: order of first appearance among the not yet adjusted patches (Chisel.cpp:199-214)
    seen = []
    for p in range(len(fid)):
        if adj[p]:
            assert cl[p] == -1 and adj2[p] == 1
            continue
        if fid[p] not in seen:
            seen.append(fid[p])
        assert cl[p] == seen.index(fid[p])
        assert adj2[p] == 1
    # untouched entries stay NaN (the binding pre-fills with NaN): adjusted and wrong-mapped patches
    for p in range(len(fid)):
        sl = slice(voff[p], voff[p + 1])
        assert np.isnan(labs[sl]).all() == bool(adj[p] or wrong[p])
    # the compensated colours of a cluster have the mesh colours' mean (exactly the construction)
    for c in range(len(seen)):
        idx = np.concatenate([np.arange(voff[p], voff[p + 1]) for p in range(len(fid)) if cl[p] == c and not wrong[p]])
        assert np.allclose(labs[idx].mean(0), mesh[idx].mean(0), atol=1e-5)


def test_oracle_all_wrong_cluster_is_left_untouched():
    fid, wrong, adj, voff, tex, mesh = _batch(2, n_patches=6, frames=(1, 2), wrong_every=2)
    wrong[:] = 0
    wrong[fid == 2] = 1  # every patch of frame 2 maps wrongly: color_src empty -> `continue` (:242)
    labs, adj2, T, cl = O.color_compensate(fid, wrong, adj, voff, tex, mesh)
    assert (adj2[fid == 1] == 1).all() and (adj2[fid == 2] == 0).all()
