"""Rank process for tests/test_launch.py: HIP-free stand-in for a bench.py rank.

Same skeleton as a GPU rank (launch.init_process_group -> forward of this rank's shard -> dp gather -> rank 0
prints ONE JSON line), with gloo instead of nccl and the CPU oracle instead of the HIP engine (this file lives
under tests/, the only place besides smoke() and bench.py's cpu_baseline that may use oracle/).
argv: n_images [fail_rank [corrupt_rank]]   (corrupt_rank >= 0: also run dp.verify_gather on a clean gather and on one whose copy on
      that rank has one slot changed; the all-reduced verdicts go into rank 0's line)
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n_images = int(sys.argv[1])
    fail_rank = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    corrupt_rank = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    assert "vision-transformer-opencl_amd.binding" not in sys.modules  # the launcher path must not need the HIP library
    rank, local, world = pkg.launch.init_process_group("gloo")
    if rank == fail_rank:
        sys.exit(7)                      # a dying rank: the launcher has to stop the others and report 7
    import torch.distributed as dist
    from oracle import pyoracle as po
    cfg = pkg.VIT_TINY
    ocfg = po.Config(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.num_classes, cfg.embed_dim, cfg.depth,
                     cfg.num_heads, cfg.hidden_dim)
    W = pkg.synth.make_weights(cfg, 5)
    imgs = pkg.synth.make_images(cfg, n_images, 6)
    local_probs, labels, probs = pkg.dp.forward_sharded(lambda x: po.forward(ocfg, x, W), imgs, rank, world)
    verify = None
    if corrupt_rank >= 0:
        import torch
        packed = pkg.dp.pack_top1(torch.arange(3, dtype=torch.int32) + 10 * rank, torch.full((3,), 0.125 * (rank + 1)))
        gathered = pkg.dp.gather_packed(packed)
        bad = gathered.clone()
        if rank == corrupt_rank:
            bad[(corrupt_rank + 3) % world, 1, 0] += 1     # this rank's copy of ANOTHER rank's slot
        verify = [bool(pkg.dp.verify_gather(packed, gathered)), bool(pkg.dp.verify_gather(packed, bad))]
    dist.barrier()
    if rank != 0:
        print(f"rank {rank} done")       # must NOT reach the parent's stdout
    else:
        print(json.dumps({"world": world, "labels": labels.tolist(), "probs": [float(p) for p in probs],
                          "n_local": int(local_probs.shape[0]), "verify": verify}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
