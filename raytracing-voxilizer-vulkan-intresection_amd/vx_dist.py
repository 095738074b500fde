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


class Exchange:
    """The one-shot exchange of a sharded build, set up once per (grid size, world): rank r's words [wb, we) of `mask` are valid
    on entry, the whole mask on exit, on every rank.

    * num_words % world == 0 (1024^3 / 8, 512^3 / any power of two): IN-PLACE all-gather -- the send buffer is the rank's own
      slice of the mask and the receive buffer the mask itself; no staging buffer, no extra device copy.
    * otherwise: the shards are padded to `chunk` words through a persistent scratch (one copy in, one copy out).
    """

    def __init__(self, num_words, rank, world, device, dist):
        import voxhip
        self.n, self.rank, self.world, self.dist = num_words, rank, world, dist
        self.wb, self.we, self.chunk = voxhip.shard_words(num_words, rank, world)
        self.inplace = num_words % world == 0 and num_words > 0
        self.bytes_per_rank = self.chunk * 4
        self.backend = dist.get_backend()
        self.algo = ("all_gather_into_tensor, in place" if self.inplace else "all_gather_into_tensor via padded scratch") + " (%s)" % self.backend
        self.scratch = None
        if not self.inplace:
            dev = device if self.backend == "nccl" else "cpu"
            self.send = torch.zeros(self.chunk, dtype=torch.int32, device=dev)
            self.scratch = torch.empty(self.chunk * world, dtype=torch.int32, device=dev)

    def run(self, mask):
        dist = self.dist
        if mask.is_cuda and self.backend != "nccl":
            # rehearsal on a box without RCCL peers (gloo): stage through host memory
            host = mask.cpu()
            exchange_bitmask(host, torch.empty(self.chunk * self.world, dtype=mask.dtype), self.wb, self.we, self.chunk, dist)
            mask.copy_(host.to(mask.device))
            return mask
        if self.inplace:
            if self.backend == "nccl":
                dist.all_gather_into_tensor(mask, mask[self.wb:self.we])   # RCCL over xGMI, in place: the production path
            else:
                _gather_fallback(mask, mask[self.wb:self.we].clone(), self.chunk, self.world, dist)
            return mask
        if self.we > self.wb:
            self.send[: self.we - self.wb] = mask[self.wb:self.we]
        if self.backend == "nccl":
            dist.all_gather_into_tensor(self.scratch, self.send)
        else:
            _gather_fallback(self.scratch, self.send, self.chunk, self.world, dist)
        mask.copy_(self.scratch[: self.n])
        return mask


def exchange_bitmask(mask, gathered, wb, we, chunk, dist):
    """mask: int32[num_words] on this rank, valid in [wb, we), anything elsewhere.  On return every rank holds the full
    mask.  gathered: int32[chunk*world] scratch."""
    world = dist.get_world_size()
    n = mask.numel()
    send = torch.zeros(chunk, dtype=mask.dtype, device=mask.device)
    if we > wb:
        send[: we - wb] = mask[wb:we]
    if mask.is_cuda and dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(gathered, send)          # RCCL over xGMI
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
