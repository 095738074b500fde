"""world_size-2 (and 3) gloo tests of the multi-GPU exchange on CPU tensors: word-disjoint shards, all-gather == OR,
and the literal all-reduce form agrees.  The shard contents come from the oracle's full mask, masked per rank -- exactly
what vx_voxelize_opts.word_begin/word_end produces on the GPU (checked there by test_word_shards_or_to_full_mask)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, full_path, out_dir):
    for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import voxhip
    import vx_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(np.load(full_path))
    n = full.numel()
    wb, we, chunk = voxhip.shard_words(n, rank, world)
    # this rank's partial mask: its own words, garbage elsewhere (must be ignored by the exchange)
    mask = torch.full((n,), -1, dtype=torch.int32)
    mask[wb:we] = full[wb:we]
    gathered = torch.empty(chunk * world, dtype=torch.int32)
    got = vx_dist.exchange_bitmask(mask.clone(), gathered, wb, we, chunk, dist)
    got2 = vx_dist.exchange_bitmask_allreduce(mask.clone(), wb, we, dist)
    # the bench's exchange object: in place when the word count divides by the world size, padded scratch otherwise
    ex = vx_dist.Exchange(n, rank, world, "cpu", dist)
    got3 = ex.run(mask.clone())
    m = n - n % world    # a divisible prefix exercises the in-place branch for every world size
    ex2 = vx_dist.Exchange(m, rank, world, "cpu", dist)
    part = torch.full((m,), -1, dtype=torch.int32)
    part[ex2.wb:ex2.we] = full[ex2.wb:ex2.we]
    got4 = ex2.run(part)
    assert ex2.inplace and ex2.bytes_per_rank == 4 * (m // world)
    ok = bool(torch.equal(got, full)) and bool(torch.equal(got2, full)) and bool(torch.equal(got3, full)) and bool(torch.equal(got4, full[:m]))
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([ok]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_bitmask_gloo(world, tmp_path, vx):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import vx_scenes
    v, t = vx_scenes.scene("adversarial")
    w, _, gi = oracle.build_bool(v, t, np.float32(0.05))   # 20^3 -> 250 words: not a multiple of the world sizes
    assert w.size % world != 0 or world == 2
    full_path = str(tmp_path / "full.npy")
    np.save(full_path, w.view(np.int32))
    mp.spawn(_worker, args=(world, _free_port(), full_path, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(str(tmp_path / ("ok%d.npy" % r)))[0]
