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
def _sweep():
    pts = []
    import re
    for line in open(os.path.join(ROOT, "profiles", "r02_density_sweep.jsonl")):
        d = json.loads(line)
        m = re.search(r"\(keep ([0-9.]+)\)", d["config"]["workload"])
        r = d.get("roofline") or {}
        pts.append((float(m.group(1)) if m else 0.62, d["ms_per_step"], r.get("avg_us_per_block")))
    return sorted(pts)


def test_mi355x_model_reproduces_the_measured_density_sweep():
    """Step time of LAUD-ResNet101 bs256 and the chained stage-3 block time at seven keep probabilities, measured on MI355X
    (profiles/r02_density_sweep.jsonl): the calibrated model within 5 % / 6 %; predicted speedup over the same kernels at density 1
    within 5 % of the realised one."""
    from laudnet_amd.predictor import BlockShape, Calibration, Predictor
    P = Predictor()
    assert P.cal.source.endswith("r02_predictor_calibration.json")
    pts = _sweep()
    assert len(pts) >= 6 and pts[-1][0] == 1.0
    stage3 = BlockShape(1024, 256, 1024, 14, 14, 1, False, 2)
    dense_ms = pts[-1][1]
    dense_pred = P.predict_resnet(256, density=(1.0,) * 4)["ms"]
    for keep, ms, chain_us in pts:
        pred = P.predict_resnet(256, density=(keep,) * 4)["ms"]
        assert abs(pred / ms - 1) < 0.05, (keep, pred, ms)
        if chain_us:
            assert abs(P.fused_block(stage3, 256, keep, True)["s"] * 1e6 / chain_us - 1) < 0.06, keep
        assert abs((dense_pred / pred) / (dense_ms / ms) - 1) < 0.05, keep
    # uncalibrated defaults must still be in the right region (the constants are physical, not free-form)
    P0 = Predictor(cal=Calibration())
    assert 0.6 < P0.predict_resnet(256)["ms"] / P.predict_resnet(256)["ms"] < 1.6


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
