"""Data-parallel sharding + top-1 gather over torch.distributed (gloo, world_size 2, CPU).

The forward of each rank is the CPU oracle on a reduced model here (no GPU in this suite); on the
GPU box bench.py drives the same helpers with the HIP engine and the nccl (RCCL) backend.
"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_images, out_dir):
    import importlib
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = pkg.VIT_TINY
    ocfg = po.Config(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.num_classes, cfg.embed_dim, cfg.depth,
                     cfg.num_heads, cfg.hidden_dim)
    W = pkg.synth.make_weights(cfg, 5)
    imgs = pkg.synth.make_images(cfg, n_images, 6)
    local, labels, probs = pkg.dp.forward_sharded(lambda x: po.forward(ocfg, x, W), imgs, rank, world)
    lo, hi = pkg.dp.shard_range(n_images, rank, world)
    # the self-check bench.py runs on its RCCL gather (dp.verify_gather), here over gloo: right as it stands; wrong -- and reported
    # wrong on BOTH ranks -- when one slot of one rank's buffer is corrupted, or when the known first labels are not there
    import torch
    packed = pkg.dp.pack_top1(torch.arange(4, dtype=torch.int32) + 10 * rank, torch.full((4,), 0.25 * (rank + 1)))
    gathered = pkg.dp.gather_packed(packed)
    good = pkg.dp.verify_gather(packed, gathered)
    bad_copy = gathered.clone()
    if rank == 1:
        bad_copy[0, 0, 2] += 1
    bad = pkg.dp.verify_gather(packed, bad_copy)
    first = pkg.dp.verify_gather(packed, gathered, expect_first_labels=[0, 1])   # rank 1's slot starts with 10, 11
    # a buffer of the wrong shape on ONE rank: that rank must still take part in every collective of the check (or the other
    # would wait in its broadcasts for ever) and both must hear "wrong"
    shape = pkg.dp.verify_gather(packed, gathered[:, :, :3] if rank == 1 else gathered)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), local=local, labels=labels, probs=probs, lo=lo, hi=hi,
             verify=np.array([good, bad, first, shape]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [6, 5])  # even and ragged split
def test_sharded_forward_and_top1_gather(tmp_path, n_images, oracle):
    import importlib
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    world = 2
    port = 29500 + (os.getpid() % 2000) + n_images
    mp.spawn(_worker, args=(world, port, n_images, str(tmp_path)), nprocs=world, join=True)
    from conftest import oracle_config
    cfg = pkg.VIT_TINY
    W = pkg.synth.make_weights(cfg, 5)
    imgs = pkg.synth.make_images(cfg, n_images, 6)
    full = oracle.forward(oracle_config(cfg), imgs, W)
    seen = 0
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        lo, hi = int(d["lo"]), int(d["hi"])
        assert np.array_equal(d["local"], full[lo:hi])          # every rank computed exactly its slice
        assert np.array_equal(d["labels"], full.argmax(1))       # and everybody holds the whole top-1 list
        assert np.array_equal(d["probs"], full.max(1))
        assert d["verify"].tolist() == [True, False, False, False]  # the same verdicts on every rank
        seen += hi - lo
    assert seen == n_images


def test_shard_range_partitions_exactly():
    import importlib
    dp = importlib.import_module("vision-transformer-opencl_amd").dp
    for n in (0, 1, 7, 256, 16384):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(4, 2, 2)
