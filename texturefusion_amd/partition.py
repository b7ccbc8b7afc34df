"""Chunk-range partition of one volume over N ranks.

Ownership is a contiguous slab of ChunkID.x per rank: rank r owns lo_r <= id.x < hi_r.  Every
rank runs the (deterministic, cheap) visible-chunk selection in full, so all ranks hold the
identical reference-ordered list; integrate / finalize then touch only owned entries
(tf_set_partition).  The only data exchanged is the payload of updated chunks in the ghost band of a
slab -- key == hi-1 and lo <= key <= lo + sum(axis): the mesher of a chunk c reads c + {0,1}^3
(Structure/ChunkManager.cpp:618-632) and the face neighbours of those (gradients, :288-315) --
all-gathered so that neighbour reads of the next stage see current TSDFs.

Host logic only (numpy); the collectives themselves are torch.distributed calls in the caller.
"""
from __future__ import annotations

import math

import numpy as np

INT_MIN = -(1 << 31)
INT_MAX = (1 << 31) - 1


def room_extent_chunks(res, half_x: float = 2.0, margin: float = 0.3):
    """ChunkID.x extent [lo, hi) of the S-room scene (walls at |x| = half_x) incl. the TSDF band."""
    edge = 8.0 * float(res)
    lo = int(math.floor((-half_x - margin) / edge))
    hi = int(math.ceil((half_x + margin) / edge)) + 1
    return lo, hi


def slab_bounds(extent, world: int):
    """Equal-width contiguous slabs; the outermost slabs extend to +-infinity so that every chunk
    id has exactly one owner."""
    lo, hi = extent
    edges = [lo + (hi - lo) * r // world for r in range(world + 1)]
    edges[0], edges[-1] = INT_MIN, INT_MAX
    return edges


def slab_for_rank(extent, rank: int, world: int):
    e = slab_bounds(extent, world)
    return e[rank], e[rank + 1]


def owner_of(ids, extent, world: int):
    """Rank owning each chunk id (ids: int32 [n,3])."""
    e = np.asarray(slab_bounds(extent, world)[1:-1], np.int64)
    x = np.asarray(ids, np.int64).reshape(-1, 3)[:, 0]
    return np.searchsorted(e, x, side="right").astype(np.int32)


def boundary_mask(ids, lo: int, hi: int, axis=(1, 0, 0)):
    """Entries of an id list that sit in the ghost band of the slab [lo, hi) of key axis . id."""
    k = np.asarray(ids, np.int64).reshape(-1, 3) @ np.asarray(axis, np.int64)
    return ((k >= lo) & (k <= lo + int(sum(axis)))) | (k == hi - 1)


def merge_needs(needs_per_rank, ids, extent, world: int):
    """needsUpdate flags of the full reference-ordered list from the per-rank flags: every rank
    only sets flags of entries it owns, so the merge is an OR; returned in list order, which makes
    validChunks / dirty lists reproduce the single-process order."""
    out = np.zeros(len(ids), np.uint8)
    own = owner_of(ids, extent, world)
    for r, nd in enumerate(needs_per_rank):
        nd = np.asarray(nd, np.uint8)
        out |= np.where(own == r, nd, 0).astype(np.uint8)
    return out


# ---- general ownership key ------------------------------------------------------------------
# key(id) = axis . id with axis in {0,1}^3.  (1,0,0) are the x slabs above; (1,1,1) cuts every
# axis-aligned wall / floor diagonally, so that a rank never holds a whole wall on its own.

def key_of(ids, axis=(1, 0, 0)):
    ids = np.asarray(ids, np.int64).reshape(-1, 3)
    return ids @ np.asarray(axis, np.int64)


def balanced_edges(keys, world: int):
    """Slab edges [INT_MIN, e_1, ..., e_{world-1}, INT_MAX] that split the given sample of chunk keys
    (e.g. the selections of a few frames spread over the stream) into equally populated slabs.
    Deterministic; edges are strictly increasing even for degenerate samples."""
    keys = np.sort(np.asarray(keys, np.int64).ravel())
    edges = [INT_MIN]
    if len(keys) == 0:
        keys = np.arange(world, dtype=np.int64)
    for r in range(1, world):
        e = int(keys[min(len(keys) - 1, (len(keys) * r) // world)])
        edges.append(max(e, edges[-1] + 1 if edges[-1] != INT_MIN else e))
    edges.append(INT_MAX)
    return edges


def owner_of_key(ids, edges, axis=(1, 0, 0)):
    e = np.asarray(edges[1:-1], np.int64)
    return np.searchsorted(e, key_of(ids, axis), side="right").astype(np.int32)


# ---- the sized neighbour exchange (host twin of FrameCtl::band_cnt / xchg_bucket, tf_volume.h) ------------------
def band_counts(ids, lo: int, hi: int, axis=(1, 0, 0)):
    """Selected chunks of a frame per ghost band, from the FULL selection every rank holds: (own down band, own up band,
    the up band of the rank below, the down band of the rank above).  Rank r's first number is what rank r - 1 computes
    as its fourth, and so on: both sides of a transfer size it alike without talking to each other."""
    k = key_of(ids, axis)
    w = int(sum(axis))
    return (int(((k >= lo) & (k - lo <= w)).sum()), int((k == hi - 1).sum()), int((k == lo - 1).sum()),
            int(((k >= hi) & (k - hi <= w)).sum()))


def xchg_bucket(count: int, cap: int = 0) -> int:
    """Record capacity of a sized block: multiples of 8, at least 8, at most cap (cap > 0)."""
    b = max(8, (int(count) + 7) // 8 * 8)
    return min(b, cap) if cap > 0 else b
