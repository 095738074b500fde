"""Multi-GPU exchange of the occupancy bitmask (one process per GPU, torch.distributed: backend "nccl" == RCCL over
xGMI on ROCm, "gloo" on CPU for the tests).

Partition: rank r owns bitmask words [wb, we) = vx_shard_words(num_words, r, world) and voxelizes only the voxels of
those words (vx_voxelize_opts.word_begin/word_end), so the ranks' contributions are WORD-DISJOINT.  The OR of the
partial masks is therefore the concatenation of the shards: one all-gather of `chunk` words per rank moves half the bytes
of an all-reduce and needs no OR operator (RCCL has none).  `exchange_bitmask_allreduce` is the literal "all-reduce"
form (sum of zero-padded masks == OR because the supports are disjoint), kept for A/B and for backends without
all_gather_into_tensor.
"""
import torch


def exchange_bitmask(mask, gathered, wb, we, chunk, dist):
    """mask: int32[num_words] on this rank, valid in [wb, we), anything elsewhere.  On return every rank holds the full
    mask.  gathered: int32[chunk*world] scratch."""
    world = dist.get_world_size()
    n = mask.numel()
    send = torch.zeros(chunk, dtype=mask.dtype, device=mask.device)
    if we > wb:
        send[: we - wb] = mask[wb:we]
    if mask.is_cuda and dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(gathered, send)          # RCCL over xGMI: the production path
        mask.copy_(gathered[:n])
    elif mask.is_cuda:
        # rehearsal on a box without RCCL peers (gloo): stage through host memory
        hs, hg = send.cpu(), torch.empty(chunk * world, dtype=mask.dtype)
        _gather_fallback(hg, hs, chunk, world, dist)
        mask.copy_(hg[:n].to(mask.device))
    else:
        _gather_fallback(gathered, send, chunk, world, dist)
        mask.copy_(gathered[:n])
    return mask


def _gather_fallback(gathered, send, chunk, world, dist):
    parts = [gathered[r * chunk:(r + 1) * chunk] for r in range(world)]
    dist.all_gather(parts, send)


def exchange_bitmask_allreduce(mask, wb, we, dist):
    """All-reduce form: zero everything outside the own shard, then SUM (== OR on disjoint supports)."""
    n = mask.numel()
    if wb > 0:
        mask[:wb].zero_()
    if we < n:
        mask[we:].zero_()
    dist.all_reduce(mask, op=dist.ReduceOp.SUM)
    return mask
