"""Data-parallel inference over the 8 GPUs of one node: one process per GPU, batch sharded
contiguously, weights replicated.  Images are independent in eval mode (BN running stats, per-image
masks), so the data path has no collective; the only exchange per batch is

  1. all_gather of the logits  [B_local,1000] fp32  -> [B_global,1000] on every rank, and
  2. one all_reduce(SUM) of a packed statistics vector (per-block sparsities, flops_perc, flops)

so that the returned 7-tuple equals what a single device would report for the global batch when shards
are equal-sized (the reference averages per-rank means the same way, train/main.py:673-683).
Backend "nccl" is RCCL on ROCm (xGMI); the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(global_batch: int, rank: int, world: int):
    """Contiguous equal shards; the global batch must divide evenly (weak scaling keeps it so)."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def _pack_stats(outputs):
    _, s3, s2, s1, cs, perc, flops = outputs
    parts = [t.reshape(-1).float() for group in (s3, s2, s1, cs) for t in group]
    parts += [perc.reshape(-1).float(), flops.reshape(-1).float()]
    return torch.cat(parts), [p.numel() for p in parts]


def _unpack_stats(vec, sizes, n_stages):
    chunks = list(torch.split(vec, sizes))
    groups = [chunks[i * n_stages:(i + 1) * n_stages] for i in range(4)]
    perc, flops = chunks[4 * n_stages], chunks[4 * n_stages + 1].reshape(())
    return groups, perc, flops


def gather_outputs(outputs, group=None):
    """(logits_local, s3[4], s2[4], s1[4], cs[4], flops_perc, flops) -> the same tuple for the GLOBAL batch."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return outputs
    world = dist.get_world_size(group)
    logits = outputs[0].contiguous()
    full = torch.empty((world * logits.shape[0],) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    dist.all_gather_into_tensor(full, logits, group=group)
    vec, sizes = _pack_stats(outputs)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    vec = vec / world
    groups, perc, flops = _unpack_stats(vec, sizes, len(outputs[1]))
    return (full, list(groups[0]), list(groups[1]), list(groups[2]), list(groups[3]), perc, flops)
