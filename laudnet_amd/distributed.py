"""Data-parallel inference over the 8 GPUs of one node: one process per GPU, batch sharded
contiguously, weights replicated.  Images are independent in eval mode (BN running stats, per-image
masks), so the data path has no collective; the only exchange per batch is

  1. all_gather of the logits  [B_local,1000] fp32  -> [B_global,1000] on every rank, and
  2. one all_reduce(SUM) of the packed per-block sparsities (each is a mean over the rank's shard;
     shards are equal-sized, so SUM / world is the global-batch mean, i.e. sum(mask) / count over the
     global batch),

after which flops_perc / flops are RECOMPUTED from the global sparsities with the model's own
shape-only FLOPs table (`model.flops_from_sparsities`): they contain channel_sparsity**2
(laud_resnet.py:129), so averaging per-rank values would not give the single-device result.  The
returned 7-tuple therefore EQUALS what one device reports for the global batch (tests/test_distributed.py
asserts 1e-6).  (The reference itself averages per-rank values, train/main.py:673-683.)

Both collectives are issued asynchronously on the backend's own stream (`gather_outputs_async`), so the
exchange of batch i overlaps the forward of batch i+1; `wait()` joins it.
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


def _pack_sparsities(outputs):
    _, s3, s2, s1, cs, _, _ = outputs
    parts = [t.reshape(-1).float() for group in (s3, s2, s1, cs) for t in group]
    return torch.cat(parts), [p.numel() for p in parts]


class PendingGather:
    """Handle of an exchange in flight (gather_outputs_async).  wait() -> the global-batch 7-tuple."""

    def __init__(self, outputs, full, vec, sizes, works, world, recompute, pf=None):
        self._o, self._full, self._vec, self._sizes = outputs, full, vec, sizes
        self._works, self._world, self._recompute, self._pf = works, world, recompute, pf

    def wait(self):
        for w in self._works:
            w.wait()          # NCCL/RCCL: the current stream waits for the collective's stream; gloo: host wait
        n_stages = len(self._o[1])
        chunks = list(torch.split(self._vec / self._world, self._sizes))
        groups = [chunks[i * n_stages:(i + 1) * n_stages] for i in range(4)]
        if self._recompute is not None:
            perc, flops = self._recompute(*groups)
        else:
            # no FLOPs table given: average the per-rank values like the reference does (train/main.py:673-683); NOT equal to
            # the single-device result when channel sparsities differ between shards
            # (that all_reduce was issued with the others by gather_outputs_async, on the same process group)
            pf = self._pf / self._world
            perc, flops = pf[:-1], pf[-1].reshape(())
        return (self._full, list(groups[0]), list(groups[1]), list(groups[2]), list(groups[3]), perc, flops)


class _Ready:
    def __init__(self, outputs):
        self._o = outputs

    def wait(self):
        return self._o


def gather_outputs_async(outputs, recompute=None, group=None):
    """Start the exchange for (logits_local, s3[4], s2[4], s1[4], cs[4], flops_perc, flops); returns a handle whose wait()
    yields the same tuple for the GLOBAL batch.  `recompute(s3, s2, s1, cs) -> (flops_perc, flops)` is the model's
    `flops_from_sparsities` bound to the input shape (see module docstring)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return _Ready(outputs)
    world = dist.get_world_size(group)
    logits = outputs[0].contiguous()
    full = torch.empty((world * logits.shape[0],) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    vec, sizes = _pack_sparsities(outputs)
    works = [dist.all_gather_into_tensor(full, logits, group=group, async_op=True),
             dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group, async_op=True)]
    pf = None
    if recompute is None:   # no FLOPs table: the per-rank flops_perc / flops are averaged (third async collective, same group)
        pf = torch.cat([outputs[5].reshape(-1).float(), outputs[6].reshape(-1).float()])
        works.append(dist.all_reduce(pf, op=dist.ReduceOp.SUM, group=group, async_op=True))
    return PendingGather(outputs, full, vec, sizes, works, world, recompute, pf)


def gather_outputs(outputs, recompute=None, group=None):
    """Blocking form of gather_outputs_async."""
    return gather_outputs_async(outputs, recompute, group).wait()


def recompute_for(model, x_shape):
    """The `recompute` callable for a laudnet_amd model and an input shape [B_local, C, H, W] (the per-image FLOPs table does
    not depend on the batch size)."""
    return lambda s3, s2, s1, cs: model.flops_from_sparsities(tuple(x_shape), s3, s2, s1, cs)


def broadcast_state(module, src: int = 0, group=None):
    """Every parameter and buffer of `module` becomes rank `src`'s (replicated weights really are replicas: e.g. maskers calibrated
    per rank on different shards).  The reference loads ONE checkpoint on every rank (train/main.py:187).  Prepared (folded / split)
    weight caches of the modules are dropped so that the next forward re-derives them from the broadcast tensors."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                dist.broadcast(t.data, src=src, group=group)
    for m in module.modules():
        drop = getattr(m, "_drop_cache", None)
        if callable(drop):
            drop()


def ranks_seen(device=None, group=None) -> int:
    """all_reduce(SUM) of a one: how many ranks the backend's collective really spans (the JSON line of bench.py carries it)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, device=device if device is not None else "cpu", dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(round(float(t.item())))


def gpu_numa_node(local_rank: int):
    """NUMA node of the GPU this rank computes on, or None when the platform does not say (single-node hosts report -1).  Resolved from the
    device's PCI bus id as the HIP runtime reports it (torch.cuda.get_device_properties(i).pci_bus_id -> /sys/bus/pci/devices/<bdf>/numa_node):
    the order of /sys/class/drm/card* is neither numeric under sorted() (card10 < card2) nor guaranteed to be HIP's device order.
    Without a visible device (CPU tests) the DRM cards are walked in NUMERIC order as a best effort."""
    try:
        if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
            pr = torch.cuda.get_device_properties(local_rank)       # (HIP_VISIBLE_DEVICES is already applied: ordinal local_rank IS this rank's GPU)
            bdf = getattr(pr, "pci_bus_id", None)
            dom = getattr(pr, "pci_domain_id", 0) or 0
            dev_id = getattr(pr, "pci_device_id", 0) or 0
            if isinstance(bdf, int):
                bdf = f"{dom:04x}:{bdf:02x}:{dev_id:02x}.0"
            if isinstance(bdf, str) and bdf:
                if bdf.count(":") == 1:
                    bdf = "0000:" + bdf
                node = int(open(f"/sys/bus/pci/devices/{bdf.lower()}/numa_node").read().strip())
                return node if node >= 0 else None
    except (OSError, ValueError, RuntimeError, AttributeError):
        pass
    import glob
    import re
    cards = []
    paths = glob.glob("/sys/class/drm/card[0-9]*/device")
    for d in sorted(paths, key=lambda q: int(re.search(r"card(\d+)", q).group(1))):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":      # AMD
                continue
            cards.append(int(open(os.path.join(d, "numa_node")).read().strip()))
        except (OSError, ValueError):
            continue
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    idx = local_rank
    if vis:
        try:
            idx = int(vis.split(",")[local_rank])
        except (ValueError, IndexError):
            idx = local_rank
    if idx >= len(cards) or cards[idx] < 0:
        return None
    return cards[idx]


def pin_to_gpu_numa_node(local_rank: int):
    """Restrict this process to the CPUs of its GPU's NUMA node (os.sched_setaffinity): with one process per GPU the launch path of every
    rank then runs next to its device instead of wherever the scheduler put it.  Returns the node, or None if nothing was changed."""
    node = gpu_numa_node(local_rank)
    if node is None:
        return None
    try:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if allowed:
            os.sched_setaffinity(0, allowed)
            return node
    except (OSError, ValueError):
        pass
    return None
