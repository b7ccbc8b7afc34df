"""Boundary-chunk exchange between chunk-range partitions through torch.distributed.

The path's only exchange step moves fixed-capacity blocks
    [u32 record count, 12 B padding | cap records of 16 + 4096 + 4096 B]
-- the count travels in-band, so nothing has to come back to the host between packing and unpacking -- either as
ONE all-gather of every rank's block (tf_boundary_pack_block / tf_boundary_unpack_blocks, include/tf_fusion.h) or,
slabs being contiguous key ranges, as one send / receive pair with the rank below and the rank above
(tf_boundary_pack_bands: a rank receives two blocks whatever the number of ranks).  The product's own transport is RCCL
inside the library (tf_comm_init / tf_exchange_boundary, texturefusion_amd/csrc/tf_comm.cpp); this helper moves the
same blocks with whatever backend the process group has (gloo on CPU in the tests, where RCCL cannot run).
torch.distributed is plumbing here.
"""
from __future__ import annotations

RECORD_BYTES = 16 + 4096 + 4096
HEADER_BYTES = 16


def block_bytes(cap: int) -> int:
    return HEADER_BYTES + cap * RECORD_BYTES


def allgather_blocks(block, group=None):
    """block: 1-D uint8 tensor of block_bytes(cap) bytes (this rank's [count | records]).  Returns the 1-D uint8
    tensor of world * block_bytes(cap) bytes every rank ends up with, rank r's block at r * block_bytes(cap)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty(block.numel() * world, dtype=torch.uint8, device=block.device)
    dist.all_gather_into_tensor(out, block, group=group)
    return out


def neighbour_exchange(block_down, block_up, group=None):
    """The neighbour form: this rank's `block_down` (what the rank below reads as ghosts) goes to rank - 1, `block_up` to
    rank + 1; returns (from_below, from_above) -- the lower neighbour's up block and the upper neighbour's down block
    (a zero-count block where there is no neighbour).  Same shapes on every rank."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    below = torch.zeros_like(block_down)
    above = torch.zeros_like(block_up)
    ops = []
    if rank > 0:
        ops += [dist.P2POp(dist.isend, block_down, rank - 1, group), dist.P2POp(dist.irecv, below, rank - 1, group)]
    if rank + 1 < world:
        ops += [dist.P2POp(dist.isend, block_up, rank + 1, group), dist.P2POp(dist.irecv, above, rank + 1, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return below, above


def neighbour_exchange_sized(block_down, block_up, recv_below_bytes: int, recv_above_bytes: int, group=None):
    """The sized neighbour form: the blocks have whatever length the frame's selection gave them
    (tf_boundary_band_bounds -> tf_boundary_block_bytes); the receive lengths come from the same computation on THIS
    rank, so nothing is negotiated.  Returns (from_below, from_above); a zero-count header where there is no neighbour."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    below = torch.zeros(max(recv_below_bytes, HEADER_BYTES), dtype=torch.uint8, device=block_down.device)
    above = torch.zeros(max(recv_above_bytes, HEADER_BYTES), dtype=torch.uint8, device=block_up.device)
    ops = []
    if rank > 0:
        ops += [dist.P2POp(dist.isend, block_down, rank - 1, group), dist.P2POp(dist.irecv, below, rank - 1, group)]
    if rank + 1 < world:
        ops += [dist.P2POp(dist.isend, block_up, rank + 1, group), dist.P2POp(dist.irecv, above, rank + 1, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return below, above
