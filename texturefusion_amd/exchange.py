"""Boundary-chunk exchange between chunk-range partitions through torch.distributed.

The path's only collective is ONE fixed-capacity all-gather of blocks
    [u32 record count, 12 B padding | cap records of 16 + 4096 + 4096 B]
-- the count travels in-band, so nothing has to come back to the host between packing and unpacking
(tf_boundary_pack_block / tf_boundary_unpack_blocks, include/tf_fusion.h).  The product's own transport is RCCL
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
