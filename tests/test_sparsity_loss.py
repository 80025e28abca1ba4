"""laudnet_amd/sparsity_loss.py against losses computed by the reference's own utils/sparsity_loss_unify.py
(tests/golden/sparsity_loss.json, written by tests/golden/make_sparsity_loss_golden.py).  Floating point: the restatement sums
the per-block penalties as a tensor, the reference as a running Python sum -- tolerance 1e-6 relative + 1e-8 absolute."""
import json
import os

import pytest
import torch

from laudnet_amd import sparsity_loss as sl

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sparsity_loss.json")))
RTOL, ATOL = 1e-6, 1e-8


def _run(case, dev="cpu"):
    t, ep, mode = case["target"], case["epoch"], case["dyn_mode"]
    chan = [torch.tensor(c, device=dev) for c in case["chan"]]
    spat = [torch.tensor(s, device=dev) for s in case["spat"]]
    perc = torch.tensor(case["perc"], device=dev)
    flops = torch.tensor(case["flops"], device=dev)
    return {
        "SparsityCriterion_bounds": sl.SparsityCriterion_bounds(t, 100, 4.1)(ep, perc, flops),
        "SparsityCriterion": sl.SparsityCriterion(t, 100, 4.1)(ep, torch.cat(chan), perc, flops),
        "SparsityCriterion_channel_factor": sl.SparsityCriterion_channel_factor(t, 100, 4.1, 2.0, None, mode)(ep, chan, perc, flops),
        "SparsityCriterion_cs": sl.SparsityCriterion_cs(t, 100, 4.1, 0.5, 0.8, mode)(ep, chan, spat, perc, flops),
        "SparsityCriterion_cs_v2": sl.SparsityCriterion_cs_v2(t, 100, 4.1, 0.5, None, mode)(ep, chan, spat, perc, flops),
        "SparsityCriterion_channel_bounds": sl.SparsityCriterion_channel_bounds(t, 100, 4.1, 3.0)(ep, torch.cat(chan), perc, flops),
        "SparsityCriterion_channel_bounds_v2": sl.SparsityCriterion_channel_bounds_v2(t, 100, 4.1, 3.0)(ep, torch.cat(chan), perc, flops),
    }


def test_every_criterion_matches_the_reference():
    assert len(GOLD["cases"]) == 108
    for case in GOLD["cases"]:
        got = _run(case)
        for name, exp in case["expected"].items():
            val = float(torch.as_tensor(got[name]).reshape(-1)[0])
            assert abs(val - exp) <= ATOL + RTOL * abs(exp), (name, case["target"], case["epoch"], case["dyn_mode"], val, exp)


def test_list_of_scalars_and_floats():
    """The reference indexes `sparsity_list[i]`: per-block Python floats or 0-dim tensors work as well as one tensor."""
    case = GOLD["cases"][3]
    crit = sl.SparsityCriterion_bounds(case["target"], 100, 4.1)
    flops = torch.tensor(case["flops"])
    a = crit(case["epoch"], torch.tensor(case["perc"]), flops)
    b = crit(case["epoch"], [torch.tensor(v) for v in case["perc"]], flops)
    c = crit(case["epoch"], list(case["perc"]), flops)
    assert torch.allclose(a, b, rtol=RTOL, atol=ATOL) and torch.allclose(a, c, rtol=RTOL, atol=ATOL)


def test_schedule_end_points():
    crit = sl.SparsityCriterion_bounds(0.5, 100, 4.1)
    assert crit.schedule(0) == 1.0 and crit.schedule(33) < 1e-30 and crit.schedule(80) < 1e-30
    # epoch 0: the band is [target, target] -> any deviation is penalised; after a third: [0, 1] -> none is
    perc = torch.tensor([0.2, 0.9])
    full = torch.tensor([4.1 * 0.5])
    assert float(crit(0, perc, full)) == pytest.approx((0.3 ** 2 + 0.4 ** 2) / 2, rel=1e-6)
    assert float(crit(50, perc, full)) == pytest.approx(0.0, abs=1e-12)


def test_gradient_flows_to_the_sparsities():
    perc = torch.tensor([0.2, 0.9, 0.5], requires_grad=True)
    flops = (perc.sum() * 1.0).reshape(1)
    loss = sl.SparsityCriterion_bounds(0.5, 100, 4.1)(10, perc, flops)
    loss.backward()
    assert perc.grad is not None and torch.isfinite(perc.grad).all() and perc.grad.abs().sum() > 0


@pytest.mark.gpu
def test_on_device_without_sync():
    """The criterion runs on the device vectors the HIP path returns (no host round trip needed) and agrees with the golden."""
    for case in GOLD["cases"][::9]:
        got = _run(case, "cuda")
        for name, exp in case["expected"].items():
            assert got[name].is_cuda
            val = float(got[name].reshape(-1)[0])
            assert abs(val - exp) <= ATOL + 2 * RTOL * abs(exp), (name, val, exp)


@pytest.mark.gpu
def test_criterion_on_the_hip_models_tuple():
    """validate()'s use (train/main.py:636,670): criterion(epoch, flops_perc_list, flops) on the tuple of the HIP model equals the
    criterion on the tuple the reference produced for the same weights, input and masker decisions (tests/golden/full_tiny.pt)."""
    from test_hip_blocks import FULL, FULL_BUILT, _hip_model
    for name in FULL_BUILT[:4]:
        fx = FULL[name]
        if "masker_run" not in fx:
            continue
        model, x = _hip_model(fx)
        with torch.no_grad():
            got = model(x, 1.0)
        exp = fx["masker_run"]
        for epoch in (0, 10, 40):
            crit = sl.SparsityCriterion_bounds(0.5, 100, float(torch.as_tensor(exp[6]).reshape(-1)[0]) * 1.3)
            a = float(crit(epoch, got[5], got[6]).reshape(-1)[0])
            b = float(crit(epoch, torch.as_tensor(exp[5]), torch.as_tensor(exp[6])).reshape(-1)[0])
            assert abs(a - b) <= 1e-6 + 1e-5 * abs(b), (name, epoch, a, b)
