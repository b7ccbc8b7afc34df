"""Boundary-chunk exchange between chunk-range partitions.

The only collective of the path: an all-gather of the records of updated chunks that sit on a
slab face.  Counts are gathered first, then payloads padded to the largest count (RCCL over
xGMI when the tensors are device-resident and the backend is "nccl"; gloo on CPU in the tests).
torch.distributed is plumbing here; the records are produced / consumed by tf_boundary_pack /
tf_boundary_unpack (include/tf_fusion.h).
"""
from __future__ import annotations

RECORD_BYTES = 16 + 4096 + 4096


def allgather_records(send, n_records: int, group=None):
    """send: 1-D uint8 tensor of capacity cap*RECORD_BYTES holding n_records records.
    Returns [(tensor, count)] per rank (own rank included)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    cnt = torch.tensor([int(n_records)], dtype=torch.int64, device=send.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(counts)
    if m == 0:
        return [(send[:0], 0) for _ in range(world)]
    # equal-sized payloads: every rank contributes its first max(count) records (the tail of a
    # shorter contribution is padding and is never unpacked)
    part = send[: m * RECORD_BYTES]
    recv = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(recv, part, group=group)
    if send.is_cuda:
        # the consumer (tf_boundary_unpack) runs on the volume's own HIP stream: make the collective's
        # result visible to it before returning
        torch.cuda.synchronize(send.device)
    return [(recv[r], counts[r]) for r in range(world)]
