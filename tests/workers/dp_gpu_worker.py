"""Rank process for tests/test_gpu_multi.py: the data-parallel split with the HIP ENGINE as the forward.

Two ranks share the one GPU of the test box (RCCL refuses two ranks on one device, so the top-1 gather of this test runs
over gloo on host tensors; the RCCL gather itself is exercised by bench.py --spawn at world size 1).  Same skeleton as a
bench.py rank: launch.init_process_group -> forward of this rank's shard on its engine -> dp gather -> rank 0 prints one
JSON line.  argv: n_images
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n_images = int(sys.argv[1])
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    binding = importlib.import_module("vision-transformer-opencl_amd.binding")
    rank, local, world = pkg.launch.init_process_group("gloo")
    import torch.distributed as dist
    cfg = pkg.VIT_SMALL
    W = pkg.synth.make_weights(cfg, 7)
    imgs = pkg.synth.make_images(cfg, n_images, 8)
    eng = binding.Engine(cfg, max_batch=8, device=0)   # every rank on the box's single GPU
    eng.load_weights(W)
    local_probs, labels, probs = pkg.dp.forward_sharded(lambda x: eng.forward(x), imgs, rank, world)
    eng.close()
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "labels": labels.tolist(), "probs": [float(p) for p in probs],
                          "n_local": int(local_probs.shape[0])}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
