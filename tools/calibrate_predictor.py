#!/usr/bin/env python3
"""Fit the free constants of laudnet_amd/predictor.py to measurements of this repository on MI355X, and validate them OUT OF SAMPLE.

Input : profiles/<TAG>_density_sweep_{channel,spatial,layer,regnet}.jsonl -- bench.py lines (--no-legs, bs256) at seven keep
        probabilities (tools/density_sweep.sh on the GPU box): step time, per-block densities (`block_densities`) and, for the channel
        workload, the chained stage-3 launch's time per block (HIP events).
Output: profiles/<TAG>_predictor_calibration.json -- constants, fit / validation tables.
The constants are fitted on the keep probabilities FIT = {0.25, 0.5, 0.75, 1.0} only; {0.4, 0.62, 0.9} are held out and reported as
`out_of_sample` (tests/test_predictor.py asserts on them)."""
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laudnet_amd.predictor import BlockShape, Calibration, Predictor  # noqa: E402

FIT = (0.25, 0.5, 0.75, 1.0)
TAG = os.environ.get("ROUND_TAG", "r06")      # which round's sweep (profiles/<TAG>_density_sweep_*.jsonl) -> profiles/<TAG>_predictor_calibration.json
OUT = os.path.join(ROOT, "profiles", f"{TAG}_predictor_calibration.json")


def load(workload):
    pts = []
    for line in open(os.path.join(ROOT, "profiles", f"{TAG}_density_sweep_{workload}.jsonl")):
        d = json.loads(line)
        r = d.get("roofline") or {}
        pts.append(dict(keep=d["config"]["keep_probability_calibrated_to"], ms=d["ms_per_step"], bd=d["block_densities"],
                        chain_us=r.get("avg_us_per_block") if "k_chain" in r.get("kernel", "") else None,
                        flops_ratio=d["config"]["mean_block_flops_ratio"]))
    return sorted(pts, key=lambda p: p["keep"])


def predict(P, workload, p):
    if workload == "channel":
        return P.predict_resnet(256, density=(p["keep"],) * 4)["ms"]
    if workload == "regnet":
        return P.predict_regnet_layerskip(256, p["bd"]["s3"])["ms"]
    return P.predict_rows_resnet(256, p["bd"]["s3"], p["bd"]["s1"], layer_mode=(workload == "layer"))["ms"]


def main():
    data = {w: load(w) for w in ("channel", "spatial", "layer", "regnet")}
    is_fit = lambda p: any(abs(p["keep"] - k) < 1e-6 for k in FIT)
    cal = Calibration()
    P = Predictor(cal=cal)
    stage3 = BlockShape(1024, 256, 1024, 14, 14, 1, False, 2)

    # ---- channel workload, stage 1: the per-image workgroup law on the chained stage-3 block time
    def chain_res(v):
        cal.cu_mfma_eff, cal.act_hbm_eff, cal.cu_l2_bytes_per_s = v[0], v[1], v[2] * 1e9
        return [(P.fused_block(stage3, 256, p["keep"], True)["s"] * 1e6 - p["chain_us"]) / p["chain_us"] for p in data["channel"] if p["chain_us"] and is_fit(p)]
    f1 = least_squares(chain_res, [0.5, 0.7, 60.0], bounds=([0.3, 0.3, 20.0], [0.8, 1.0, 200.0]))
    chain_res(f1.x)

    # ---- channel workload, stage 2: dense-kernel efficiency and the per-forward residual on the step time
    # (+ the per-step cost of the one-launch stage-4 blocks and the penalty of their single-slot rings, keep 1.0)
    def step_res(v):
        cal.dense_mfma_eff, cal.fixed_s, cal.small_step_s, cal.small_single_slot = v[0], v[1] * 1e-3, v[2] * 1e-6, v[3]
        return [(predict(P, "channel", p) - p["ms"]) / p["ms"] for p in data["channel"] if is_fit(p)]
    f2 = least_squares(step_res, [0.28, 0.2, 0.93, 1.55], bounds=([0.15, 0.0, 0.6, 1.0], [0.5, 2.0, 1.3, 2.5]))
    step_res(f2.x)

    # ---- packed-row workloads: one set of row-kernel constants for spatial + layer (+ RegNet with its grouped-conv efficiency)
    def rows_res(v):
        cal.rows_cu_eff, cal.narrow_alpha, cal.tile_fixed_s, cal.rows_hbm_eff, cal.grouped_eff, cal.rows_fixed_s = v[0], v[1], v[2] * 1e-6, v[3], v[4], v[5] * 1e-3
        return [(predict(P, w, p) - p["ms"]) / p["ms"] for w in ("spatial", "layer", "regnet") for p in data[w] if is_fit(p)]
    f3 = least_squares(rows_res, [0.55, 0.5, 8.0, 0.5, 0.5, 0.3], bounds=([0.2, 0.0, 0.0, 0.2, 0.2, 0.0], [1.0, 1.5, 40.0, 0.9, 1.0, 2.0]))
    rows_res(f3.x)

    tables, summary = {}, {}
    for w, pts in data.items():
        rows = []
        for p in pts:
            pred = predict(P, w, p)
            row = dict(keep=p["keep"], used_for_fit=is_fit(p), flops_ratio=p["flops_ratio"], measured_ms=round(p["ms"], 3), predicted_ms=round(pred, 3),
                       rel_err=round(pred / p["ms"] - 1, 4))
            if w == "channel" and p["chain_us"]:
                ch = P.fused_block(stage3, 256, p["keep"], True)["s"] * 1e6
                row.update(measured_chain_us_per_block=round(p["chain_us"], 1), predicted_chain_us_per_block=round(ch, 1))
            rows.append(row)
        dense = rows[-1]
        for r in rows:
            r["realised_speedup_vs_keep1"] = round(dense["measured_ms"] / r["measured_ms"], 3)
            r["predicted_speedup_vs_keep1"] = round(dense["predicted_ms"] / r["predicted_ms"], 3)
        tables[w] = rows
        summary[w] = dict(max_abs_rel_err_in_sample=max(abs(r["rel_err"]) for r in rows if r["used_for_fit"]),
                          max_abs_rel_err_out_of_sample=max(abs(r["rel_err"]) for r in rows if not r["used_for_fit"]))
    consts = {k: getattr(cal, k) for k in ("cu_mfma_eff", "act_hbm_eff", "cu_l2_bytes_per_s", "phase_cost_s", "dense_mfma_eff", "hbm_eff", "co_resident_gain",
                                            "launch_s", "fixed_s", "rows_cu_eff", "narrow_alpha", "tile_fixed_s", "rows_hbm_eff", "idx_s",
                                            "grouped_eff", "rows_fixed_s", "small_step_s", "small_single_slot")}
    out = dict(constants=consts,
               fitted=["cu_mfma_eff", "act_hbm_eff", "cu_l2_bytes_per_s", "dense_mfma_eff", "fixed_s", "rows_cu_eff", "narrow_alpha", "tile_fixed_s", "rows_hbm_eff",
                       "grouped_eff", "rows_fixed_s", "small_step_s", "small_single_slot"],
               fit_keeps=list(FIT), held_out_keeps=[0.4, 0.62, 0.9],
               source=f"profiles/{TAG}_density_sweep_{{channel,spatial,layer,regnet}}.jsonl (tools/density_sweep.sh on one MI355X)",
               summary=summary, tables=tables)
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(dict(constants=consts, summary=summary), indent=1))
    for w, rows in tables.items():
        for r in rows:
            print(w, r["keep"], "fit" if r["used_for_fit"] else "HELD OUT", r["measured_ms"], r["predicted_ms"], r["rel_err"])


if __name__ == "__main__":
    main()
