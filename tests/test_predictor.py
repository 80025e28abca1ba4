"""Predictor side of the metric: tools/predict_speedup.py drives the reference's analytic latency model with
the reference's own block formulas.  The harness is pinned by reproducing, on the V100 preset, the four latencies the
reference's DyNetSimulator/eval_example.py computes -- stored in tests/golden/predictor_v100.json by
tests/golden/make_predictor_golden.py, which RUNS that script unmodified.  Needs /root/reference (build container only)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "predictor_v100.json")
needs_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/DyNetSimulator"), reason="the reference simulator is not on this box")


def test_golden_file_is_sane():
    g = json.load(open(GOLD))["models"]
    assert set(g) == {"resnet50", "resnet101", "regnety008"}
    assert g["resnet101"]["static_latency"] > g["resnet50"]["static_latency"] > g["regnety008"]["static_latency"] > 0
    for m in g.values():
        assert all(v > 0 for v in m.values())


@needs_ref
@pytest.mark.parametrize("model", ["resnet50", "resnet101", "regnety008"])
def test_harness_reproduces_reference_v100_numbers(model):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import predict_speedup as PS
    got = PS.reproduce_v100(model)
    want = json.load(open(GOLD))["models"][model]
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-12 + 1e-9 * v, (model, k, got[k], v)


def test_committed_prediction_has_uncalibrated_and_calibrated_rows():
    rows = json.load(open(os.path.join(ROOT, "profiles", "predicted_speedup_mi355x.json")))["rows"]
    assert any("channel-2222" in r["workload"] and r["mem_bandwidth"].startswith("8.0 TB/s (spec)") for r in rows)


# ---- the MI355X-native latency model (laudnet_amd/predictor.py) against the committed measurements
FIT_KEEPS = (0.25, 0.5, 0.75, 1.0)        # what tools/calibrate_predictor.py fits on; 0.4 / 0.62 / 0.9 are held out


def _sweep(workload):
    pts = []
    for line in open(os.path.join(ROOT, "profiles", f"r06_density_sweep_{workload}.jsonl")):
        d = json.loads(line)
        r = d.get("roofline") or {}
        pts.append(dict(keep=d["config"]["keep_probability_calibrated_to"], ms=d["ms_per_step"], bd=d["block_densities"],
                        chain_us=r.get("avg_us_per_block") if "k_chain" in r.get("kernel", "") else None))
    return sorted(pts, key=lambda p: p["keep"])


def _predict(P, workload, p):
    if workload == "channel":
        return P.predict_resnet(256, density=(p["keep"],) * 4)["ms"]
    if workload == "regnet":
        return P.predict_regnet_layerskip(256, p["bd"]["s3"])["ms"]
    return P.predict_rows_resnet(256, p["bd"]["s3"], p["bd"]["s1"], layer_mode=(workload == "layer"))["ms"]


# (round 6 re-fit: the held-out points 0.4 / 0.62 / 0.9 -- the operating range -- are within 5.1 %; the fit point keep 0.25 of the packed-row
# workloads is off by 9-12 %: the model's fixed per-block terms predate the fused maskers, which matter most where little else is left)
@pytest.mark.parametrize("workload,tol_fit,tol_held_out", [("channel", 0.03, 0.035), ("spatial", 0.13, 0.06), ("layer", 0.09, 0.05),
                                                           ("regnet", 0.06, 0.03)])
def test_mi355x_model_out_of_sample(workload, tol_fit, tol_held_out):
    """Step time of the four bench workloads (bs256) at seven keep probabilities, measured on MI355X
    (profiles/r06_density_sweep_*.jsonl, one gpurun call; re-measured and re-fitted in round 6 on the kernels of rounds 5-6).  The constants are fitted on keep 0.25 / 0.5 / 0.75 / 1.0 ONLY
    (tools/calibrate_predictor.py); the assertion that matters is the one on the HELD-OUT points 0.4 / 0.62 / 0.9 -- the packed-row
    workloads take the per-block densities the module reports (bench.py: block_densities) as their input."""
    from laudnet_amd.predictor import BlockShape, Predictor
    P = Predictor()
    assert P.cal.source.endswith("r06_predictor_calibration.json")
    cal = json.load(open(os.path.join(ROOT, "profiles", "r06_predictor_calibration.json")))
    assert tuple(cal["fit_keeps"]) == FIT_KEEPS
    pts = _sweep(workload)
    assert len(pts) == 7 and pts[-1]["keep"] == 1.0
    dense_ms, dense_pred = pts[-1]["ms"], _predict(P, workload, pts[-1])
    held_out = 0
    stage3 = BlockShape(1024, 256, 1024, 14, 14, 1, False, 2)
    for p in pts:
        fit = any(abs(p["keep"] - k) < 1e-6 for k in FIT_KEEPS)
        held_out += not fit
        pred = _predict(P, workload, p)
        assert abs(pred / p["ms"] - 1) < (tol_fit if fit else tol_held_out), (workload, p["keep"], pred, p["ms"])
        # the predicted speedup over the same kernels with everything kept (eval_example.py:203-216 vs :219-360 for this implementation)
        assert abs((dense_pred / pred) / (dense_ms / p["ms"]) - 1) < (0.08 if p["keep"] >= 0.4 else 0.15), (workload, p["keep"])
        if workload == "channel" and p["chain_us"]:
            assert abs(P.fused_block(stage3, 256, p["keep"], True)["s"] * 1e6 / p["chain_us"] - 1) < 0.08, p["keep"]
    assert held_out == 3


def test_mi355x_model_sees_the_tile_round_steps():
    """The measured step of the layer workload between keep 0.62 and 0.75 (+37 % time for +21 % rows: the stage-3 row kernels go
    from one round of workgroups to two) must come out of the model's rounds term, not be smoothed away."""
    from laudnet_amd.predictor import Predictor
    P = Predictor()
    pts = {round(p["keep"], 2): p for p in _sweep("layer")}
    meas = pts[0.75]["ms"] / pts[0.62]["ms"]
    pred = _predict(P, "layer", pts[0.75]) / _predict(P, "layer", pts[0.62])
    assert meas > 1.3 and abs(pred / meas - 1) < 0.06
    assert P.tile_columns(1024, 50176) == 256 and P.tile_columns(256, 50176) == 128 and P.tile_columns(256, 50176, taps=9) == 128
    assert P.tile_columns(64, 802816) == 64 and P.tile_columns(320, 50176) == 160


def test_mi355x_model_counts_tile_padding_and_prefers_coarser_groups():
    from laudnet_amd.predictor import BlockShape, Predictor, expected_padded_channels
    assert expected_padded_channels(128, 2, 1.0) == 256
    assert expected_padded_channels(128, 2, 0.0) == 0
    e = expected_padded_channels(128, 2, 0.62)
    assert 0.62 * 256 < e < 0.62 * 256 + 32            # padded to the 32-wide MFMA tile: between K and K + 32
    P = Predictor()
    lat = P.best_channel_granularity(BlockShape(1024, 256, 1024, 14, 14, 1, False, 2), 256, 0.62)
    assert set(lat) == {2, 4, 8, 16, 32} and all(v > 0 for v in lat.values())
    s = [P.predicted_speedup(256, density=(d,) * 4)["speedup"] for d in (0.9, 0.62, 0.4)]
    assert s[0] < s[1] < s[2]
