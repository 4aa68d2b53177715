"""Batch data-parallel helpers: one process per GPU, images sharded across ranks.

The reference has no multi-device code; its batch is a serial loop over
independent images (ViT_seq.c:354), so the path shards with NO data-path collective: every rank
holds a full weight replica and forwards its own contiguous slice of the batch.  The only
exchange is the one the north star names -- gathering the per-image top-1 records
(label, probability) -- done with one all-gather over torch.distributed (RCCL on GPUs via the
"nccl" backend, gloo on CPU for the tests).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of [0, n_images): rank r owns [r*n/world, (r+1)*n/world)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return (n_images * rank) // world, (n_images * (rank + 1)) // world


def pack_top1(labels, probs):
    """(labels int32 [n], probs float32 [n]) -> one int32 tensor [2][n] (probability bits in row 1)."""
    import torch
    labels = torch.as_tensor(labels, dtype=torch.int32)
    probs = torch.as_tensor(probs, dtype=torch.float32)
    return torch.stack([labels, probs.view(torch.int32)])


def unpack_top1(packed):
    import torch
    return packed[..., 0, :].contiguous(), packed[..., 1, :].contiguous().view(torch.float32)


def gather_packed(packed, out=None, group=None):
    """All-gather one packed [2][n] int32 record tensor per rank into out [world][2][n] (allocated when None).

    This is the ONE exchange of the data-parallel path (north star: "RCCL over xGMI only to gather top-1"):
    8 bytes per image, issued on the current stream of the tensor's device, so it is ordered after the forward
    that produced `packed` when both use the same stream.  nccl (= RCCL) for GPU tensors, gloo for CPU tensors.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(packed.shape), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1), packed.contiguous().view(-1), group=group)
    return out


def verify_gather(packed, gathered, expect_first_labels=None, group=None) -> bool:
    """Check the result of gather_packed on EVERY rank, with another collective than the one under test: slot r of `gathered`
    must equal rank r's own `packed` records (one broadcast per rank), and -- when every rank's batch starts with images whose
    labels are known (bench.py puts the golden images there) -- hold `expect_first_labels` in its first entries.  Returns the
    verdict that is the same on all ranks (all_reduce MIN): True only if every slot was right everywhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    shape_ok = tuple(gathered.shape) == (world,) + tuple(packed.shape)
    ok = shape_ok
    # every rank runs every broadcast whatever it has found so far (a rank that skipped them would leave the others waiting in
    # theirs); only the comparisons depend on the local shape, and the all_reduce below carries the verdict to everybody
    for r in range(world):
        theirs = packed.clone()
        dist.broadcast(theirs, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
        if not shape_ok:
            continue
        ok = ok and bool(torch.equal(gathered[r], theirs))
        if expect_first_labels is not None:
            k = len(expect_first_labels)
            ok = ok and gathered[r, 0, :k].cpu().tolist() == [int(v) for v in expect_first_labels]
    flag = torch.tensor([1.0 if ok else 0.0], device=packed.device, dtype=torch.float32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def gather_top1(labels, probs, counts=None, group=None):
    """All-gather the per-image top-1 records of every rank, in rank (= image) order.

    `counts[r]` = images of rank r (ragged shards are padded to the largest and trimmed after).
    Works on whatever device the tensors live on (nccl for GPU tensors, gloo for CPU).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    packed = pack_top1(labels, probs)
    n_local = packed.shape[1]
    if counts is None:
        counts = [n_local] * world
    width = max(counts)
    if n_local < width:
        packed = torch.nn.functional.pad(packed, (0, width - n_local))
    out = gather_packed(packed, None, group)
    lab, pr = [], []
    for r in range(world):
        l, p = unpack_top1(out[r])
        lab.append(l[:counts[r]])
        pr.append(p[:counts[r]])
    return torch.cat(lab), torch.cat(pr)


def forward_sharded(forward_fn: Callable[[np.ndarray], np.ndarray], images: np.ndarray, rank: int, world: int,
                    group=None):
    """Each rank forwards its slice of `images` with `forward_fn` (-> probs [n_local][classes]) and
    every rank receives the top-1 (label, prob) of the whole batch."""
    import torch
    lo, hi = shard_range(images.shape[0], rank, world)
    local = forward_fn(images[lo:hi]) if hi > lo else np.zeros((0, 1), np.float32)
    labels = local.argmax(1).astype(np.int32) if hi > lo else np.zeros(0, np.int32)
    probs = local.max(1).astype(np.float32) if hi > lo else np.zeros(0, np.float32)
    counts = [shard_range(images.shape[0], r, world)[1] - shard_range(images.shape[0], r, world)[0]
              for r in range(world)]
    all_l, all_p = gather_top1(torch.from_numpy(labels), torch.from_numpy(probs), counts, group)
    return local, all_l.numpy(), all_p.numpy()
