"""Chisel::DrawMeshes vertex / index packing (Structure/Chisel.cpp:288-355):
a hand-computed known answer for the oracle restatement (CPU); the HIP path is compared against it, bit-exact,
in tests/test_gpu_atlas.py."""
import numpy as np
import pytest

from oracle import api as O

AW = AH = 13824


def _batch(seed, n_patches=50):
    rng = np.random.default_rng(seed)
    nv = rng.integers(3, 90, n_patches)
    ni = 3 * rng.integers(1, 60, n_patches)
    voff = np.concatenate([[0], np.cumsum(nv)]).astype(np.int64)
    ioff = np.concatenate([[0], np.cumsum(ni)]).astype(np.int64)
    complete = (rng.random(n_patches) > 0.2).astype(np.uint8)
    wrong = (rng.random(n_patches) > 0.8).astype(np.uint8)
    labs_valid = ((rng.random(n_patches) > 0.3) & (wrong == 0)).astype(np.uint8)
    slot = rng.integers(0, 576 * 700, n_patches)
    texloc = ((slot // 576) * 18 * AW + (slot % 576) * 24).astype(np.uint64)
    ratio = np.where(rng.random((n_patches, 2)) > 0.7, rng.random((n_patches, 2)) * 0.9 + 0.05, 1.0).astype(np.float32)
    N = int(voff[-1])
    verts = (rng.random((N, 3)) * 4 - 2).astype(np.float32)
    colors = rng.random((N, 3)).astype(np.float32)
    normals = rng.normal(size=(N, 3)).astype(np.float32)
    texcoord = (rng.random((N, 2)) * 40).astype(np.float32)
    texcolor = rng.random((N, 3)).astype(np.float32)
    labs = np.clip(texcolor + rng.normal(scale=0.2, size=(N, 3)), -0.5, 1.5).astype(np.float32)
    indices = np.concatenate([rng.integers(0, nv[p], ni[p]) for p in range(n_patches)]).astype(np.uint32)
    return dict(complete=complete, wrong_mapping=wrong, labs_valid=labs_valid, texloc=texloc, ratio=ratio, voff=voff,
                verts=verts, colors=colors, normals=normals, texcoord=texcoord, texcolor=texcolor, labs=labs,
                ioff=ioff, indices=indices)


def test_oracle_known_answer():
    """One vertex by hand: colour (0.5, 0.25, 1.0) -> 127<<16 | 63<<8 | 255; labs - texcolor = (0.1, -0.1, 0)
    -> ((25+255)<<18) + ((-25+255)<<9) + 255; slot at texel (48, 36), ratio (0.5, 1), texcoord (10, 7)."""
    tl = np.array([36 * AW + 48], np.uint64)
    v, i = O.pack_vertices([1], [0], [1], tl, [[0.5, 1.0]], AW, AH, [0, 1], [[1, 2, 3]], [[0.5, 0.25, 1.0]],
                           [[0, 0, 1]], [[10, 7]], [[0.4, 0.5, 0.5]], [[0.5, 0.4, 0.5]], [0, 3], [0, 0, 0])
    a0 = int(np.float32(np.float32(0.5) - np.float32(0.4)) * np.float32(255)) + 255
    a1 = int(np.float32(np.float32(0.4) - np.float32(0.5)) * np.float32(255)) + 255
    want = [1, 2, 3, 50, np.float32((127 << 16) + (63 << 8) + 255), np.float32(((a0 << 9) + a1 << 9) + 255),
            np.float32(np.float32(10 * 0.5 + 48) / np.float32(AW)), np.float32(np.float32(7 + 36) / np.float32(AH)),
            0, 0, 1, 0]
    assert np.array_equal(v[0], np.array(want, np.float32))
    assert np.array_equal(i, [0, 0, 0])


def test_oracle_skips_incomplete_and_rebases_indices():
    b = _batch(3, n_patches=12)
    v, i = O.pack_vertices(atlas_w=AW, atlas_h=AH, **b)
    keep = b["complete"].astype(bool)
    nv = np.diff(b["voff"])[keep]
    assert len(v) == nv.sum() and len(i) == np.diff(b["ioff"])[keep].sum()
    # indices of the k-th complete patch point into its own vertex range
    base, pos = 0, 0
    for p in np.where(keep)[0]:
        n_i = b["ioff"][p + 1] - b["ioff"][p]
        blk = i[pos:pos + n_i]
        assert blk.min() >= base and blk.max() < base + (b["voff"][p + 1] - b["voff"][p])
        assert np.array_equal(blk - base, b["indices"][b["ioff"][p]:b["ioff"][p + 1]])
        base += b["voff"][p + 1] - b["voff"][p]
        pos += n_i
