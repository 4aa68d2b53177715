"""The rank launcher bench.py uses for `--gpus N` (vision-transformer-opencl_amd/launch.py), driven HIP-free:
world size 2 and 8, gloo, CPU oracle as the forward (tests/workers/dp_cpu_worker.py)."""
import importlib
import io
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "dp_cpu_worker.py")


@pytest.mark.parametrize("n_images", [6, 5])
def test_launcher_runs_two_gloo_ranks_and_relays_rank0(n_images, oracle):
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    sink = io.StringIO()
    rc, out = pkg.launch.launch_ranks(WORKER, [str(n_images)], 2, timeout=300, relay_stdout=sink)
    assert rc == 0
    assert out == sink.getvalue()
    assert "rank 1 done" not in out                               # rank 1's stdout is not relayed
    lines = [l for l in out.splitlines() if l.startswith("{")]   # (gloo prints a connection banner on stdout)
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    from conftest import oracle_config
    cfg = pkg.VIT_TINY
    full = oracle.forward(oracle_config(cfg), pkg.synth.make_images(cfg, n_images, 6), pkg.synth.make_weights(cfg, 5))
    assert rec["world"] == 2
    assert rec["n_local"] == pkg.dp.shard_range(n_images, 0, 2)[1]
    assert rec["labels"] == full.argmax(1).tolist()               # every image's top-1 reached rank 0, in image order
    assert np.array_equal(np.asarray(rec["probs"], np.float32), full.max(1))


def test_launcher_walks_the_eight_rank_path_with_a_ragged_split(oracle):
    """BASELINE.json configs[3] runs one rank per GPU on an 8-GPU node: the same launcher, sharding and gather at their real rank
    count (gloo, CPU oracle as the forward): 21 images over 8 ranks (rank r owns [21 r / 8, 21 (r + 1) / 8): 2,3,2,3,3,2,3,3), every image's top-1 on rank 0 in image
    order, and dp.verify_gather -- the check bench.py runs on its RCCL gather -- right as it stands and wrong on every rank when
    rank 5's copy of one slot is corrupted (the verdict is all-reduced, so rank 0's line carries it)."""
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    n_images, world = 21, 8
    assert [pkg.dp.shard_range(n_images, r, world)[1] - pkg.dp.shard_range(n_images, r, world)[0] for r in range(world)] == [2, 3, 2, 3, 3, 2, 3, 3]
    rc, out = pkg.launch.launch_ranks(WORKER, [str(n_images), "-1", "5"], world, timeout=600, relay_stdout=io.StringIO())
    assert rc == 0, out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    from conftest import oracle_config
    cfg = pkg.VIT_TINY
    full = oracle.forward(oracle_config(cfg), pkg.synth.make_images(cfg, n_images, 6), pkg.synth.make_weights(cfg, 5))
    assert rec["world"] == world and rec["n_local"] == 2
    assert rec["labels"] == full.argmax(1).tolist()
    assert np.array_equal(np.asarray(rec["probs"], np.float32), full.max(1))
    assert rec["verify"] == [True, False]


def test_launcher_reports_a_failed_rank_and_stops_the_rest():
    pkg = importlib.import_module("vision-transformer-opencl_amd")
    rc, out = pkg.launch.launch_ranks(WORKER, ["4", "1"], 2, timeout=120, relay_stdout=io.StringIO())
    assert rc == 7
    assert not [l for l in out.splitlines() if l.startswith("{")]


def test_bench_parent_spawns_before_touching_torch_or_hip():
    """`python bench.py --gpus 2` started plainly must become a launcher: the parent may not import torch or load the
    HIP library (a process that has initialised the GPU must not start workers).  It counts the node's GPUs in a child
    interpreter first and fails fast when there are fewer than ranks -- no GPU here, so that is what happens: what is checked is
    who imported what, and that nothing was started."""
    code = (
        "import sys, runpy, os\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '2', '--no-cpu-baseline']\n"
        "try:\n"
        "    runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')\n"
        "except SystemExit as e:\n"
        "    rc = e.code\n"
        "bad = [m for m in sys.modules if m == 'torch' or m.endswith('.binding')]\n"
        "print('PARENT', rc, bad)\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VIT_LAUNCH_CHILD")}
    env["HIP_VISIBLE_DEVICES"] = ""      # keep every process off any GPU: this is the CPU suite
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    last = [l for l in r.stdout.splitlines() if l.startswith("PARENT")][-1]
    assert last.endswith("[]"), r.stdout + r.stderr              # neither torch nor the binding in the parent
    assert "needs 2 GPUs on this node, it shows 0" in last and "nothing was started" in last, last
    assert "[rank" not in r.stderr, r.stderr                      # no rank ever ran


def test_a_rank_without_a_gpu_says_so_and_the_launcher_relays_it(capsys):
    """The ranks' own check (a launcher whose count could not be taken starts them anyway): `bench.py` as a launched rank on a
    box without a GPU exits with "needs a GPU", and launch_ranks relays the message and the failure."""
    from importlib import import_module
    launch = import_module("vision-transformer-opencl_amd.launch")
    rc, _ = launch.launch_ranks(os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline"],
                                2, timeout=600, extra_env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert rc != 0
    # whichever rank fails first stops the other (launch_ranks terminates the rest), so only ONE message is guaranteed
    assert re.search(r"\[rank [01]\] .*needs a GPU", capsys.readouterr().err)
