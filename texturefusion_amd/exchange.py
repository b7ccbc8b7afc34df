"""Boundary-chunk exchange between chunk-range partitions.

The only collective of the path: an all-gather of the records of updated chunks that sit on a
slab face.  Counts are gathered first, then payloads padded to the largest count (RCCL over
xGMI when the tensors are device-resident and the backend is "nccl"; gloo on CPU in the tests).
torch.distributed is plumbing here; the records are produced / consumed by tf_boundary_pack /
tf_boundary_unpack (include/tf_fusion.h).
"""
from __future__ import annotations

RECORD_BYTES = 16 + 4096 + 4096


def allgather_records(send, n_records, group=None, synchronize=True):
    """send: 1-D uint8 tensor of capacity cap*RECORD_BYTES holding n_records records; n_records is an
    int or a 1-element device tensor (tf_boundary_pack_async leaves the count on the device).
    Returns [(tensor, count)] per rank (own rank included).  With synchronize=False the caller orders
    the consumer behind the collective itself (stream events), e.g. to overlap the exchange of one
    frame batch with the integration of the next."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if torch.is_tensor(n_records):
        cnt = n_records.reshape(1).to(torch.int64)
    else:
        cnt = torch.tensor([int(n_records)], dtype=torch.int64, device=send.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    cap = send.numel() // RECORD_BYTES
    if max(counts) > cap:
        raise RuntimeError("boundary buffer too small: %d records > capacity %d" % (max(counts), cap))
    m = max(counts)
    if m == 0:
        return [(send[:0], 0) for _ in range(world)]
    # equal-sized payloads: every rank contributes its first max(count) records (the tail of a
    # shorter contribution is padding and is never unpacked)
    part = send[: m * RECORD_BYTES]
    recv = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(recv, part, group=group)
    if send.is_cuda and synchronize:
        # the consumer (tf_boundary_unpack) runs on the volume's own HIP stream: make the collective's
        # result visible to it before returning
        torch.cuda.synchronize(send.device)
    return [(recv[r], counts[r]) for r in range(world)]
