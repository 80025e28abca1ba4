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
