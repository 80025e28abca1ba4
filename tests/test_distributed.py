"""N>1 path on CPU: two gloo ranks shard a batch, run the (oracle) model on their shard and exchange results with
laudnet_amd.distributed.gather_outputs -- the same function bench.py uses over RCCL.  The gathered 7-tuple must equal
the single-process result on the full batch (logits bit-for-bit per shard order; statistics as global-batch means)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(seed=3):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from fill import fill_state_dict, seeded_randn
    from oracle import torch_ref as TR
    kw = dict(dyn_mode=["channel", "spatial", "layer", "channel"], channel_dyn_granularity=[2, 2, 2, 2],
              channel_masker_layers=[2, 2, 2, 2], mask_spatial_granularity=[2, 2, 2, 1], width_mult=0.125,
              input_size=64, num_classes=10)
    model = TR.resnet50_ref(**kw).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed))
    x = seeded_randn((4, 3, 64, 64), 11)
    return model, x


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    from laudnet_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model, x = _build()
    lo, hi = D.shard_bounds(x.shape[0], rank, world)
    with torch.no_grad():
        local = model(x[lo:hi], 1.0)
    full = D.gather_outputs(local)
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path):
    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=False)
    model, x = _build()
    with torch.no_grad():
        want = model(x, 1.0)
    assert torch.allclose(got[0], want[0], atol=1e-5), "gathered logits must equal the full-batch logits"
    for g_list, w_list in zip(got[1:5], want[1:5]):
        for g, w in zip(g_list, w_list):
            assert torch.allclose(g, w, atol=1e-6), "per-stage sparsities must be global-batch means"
    # flops_perc / flops contain channel_sparsity**2 (laud_resnet.py:129): the mean over ranks of a per-rank square is
    # not the square of the global mean.  The reference averages per-rank values the same way (train/main.py:673-683).
    assert torch.allclose(got[5], want[5], atol=2e-3)
    assert torch.allclose(got[6], want[6], rtol=2e-3)


def test_shard_bounds():
    from laudnet_amd import distributed as D
    assert [D.shard_bounds(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    with pytest.raises(ValueError):
        D.shard_bounds(10, 0, 4)
