#!/usr/bin/env python3
"""Predicted speedup half of the BASELINE metric ("realised-vs-predicted speedup").

Runs the REFERENCE's own analytic latency predictor (DyNetSimulator/hardware_models, imported unmodified from
/root/reference -- build container only, it cannot travel) with MI355X parameters and the densities bench.py
calibrates to, using the reference's block formulas (DyNetSimulator/eval_example.py:12-122, imported as a module), and
writes profiles/predicted_speedup_mi355x.json.  bench.py reads that file and reports predicted next to realised.

MI355X parameters: n_pes = 256 CUs, pe_fp32s = 128 MAC lanes per CU (256*128*2*2.4e9 = 157 TFLOP/s fp32),
frequency 2.4 GHz, memory bandwidth 8 TB/s (spec) -- and, as a second row, 6.3 TB/s (achievable, MI355X_MICROARCH.md).
The predictor's remaining knobs (l2_speed_frac, mem_concurrent_fp32, launch time) are left at the reference's defaults:
they were fitted to NVIDIA parts and are NOT calibrated for CDNA4, which is stated in the output.
"""
import contextlib
import io
import json
import os
import sys

REF = "/root/reference/DyNetSimulator"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODELS = {
    "resnet101": dict(widths=[56, 28, 14, 7], last=[256, 512, 1024, 2048], first=[64, 256, 512, 1024], strides=[1, 2, 2, 2],
                      bottleneck=4, is_se=False, group_width=None, n_block=[3, 4, 23, 3]),
    "regnety008": dict(widths=[56, 28, 14, 7], last=[64, 144, 320, 784], first=[32, 64, 144, 320], strides=[2, 2, 2, 2],
                       bottleneck=1, is_se=True, group_width=16, n_block=[1, 3, 8, 2]),
}


def run(E, predictor, model, mode, density, s_granul, c_granul):
    m = MODELS[model]
    groups = [1] * 4 if m["group_width"] is None else [c // m["group_width"] for c in m["last"]]
    total = 0.0
    for st in range(4):
        for j in range(m["n_block"][st]):
            first = j == 0
            kw = dict(c_in=m["first"][st] if first else m["last"][st], c_out=m["last"][st], b=m["bottleneck"],
                      n_groups=groups[st], h=m["widths"][st] * (m["strides"][st] if first else 1),
                      w=m["widths"][st] * (m["strides"][st] if first else 1), stride=m["strides"][st] if first else 1,
                      down=m["strides"][st] if first else 1, is_se=m["is_se"])
            if mode == "static":
                total += E.get_static_block_latency(predictor, **kw)
            elif mode == "spatial":
                total += E.get_dynamic_block_latency_spatial(predictor, granul_size=s_granul[st], c_granul_size=1,
                                                             density_conv1=density, density_conv2=density,
                                                             density_conv3=density, c_density=1.0, **kw)
            elif mode == "layer":
                total += E.get_skipping_block_latency(predictor, granul_size=m["widths"][st], c_granul_size=1,
                                                      density_conv1=density, density_conv2=density, density_conv3=density,
                                                      c_density=1, **kw)
            elif mode == "channel":
                total += E.get_dynamic_block_latency_channel(predictor, granul_size=1, c_granul_size=c_granul[st],
                                                             density_conv1=1.0, density_conv2=1.0, density_conv3=1.0,
                                                             c_density=density, layer=2, **kw)
    return float(total)


def main():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import eval_example as E                      # block formulas (functions only; __main__ guard skips the CLI)
        from hardware_models.multi_cores import GPGPUDynamicPredictor
    out = {"source": "reference DyNetSimulator imported unmodified; MI355X parameters per SURVEY 8d; reference-default "
                     "l2_speed_frac / mem_concurrent_fp32 / launch time (fitted to NVIDIA parts, not calibrated for CDNA4)",
           "batch": 256, "rows": []}
    for bw_name, bw in (("8.0 TB/s (spec)", 8.0e12), ("6.3 TB/s (achievable)", 6.3e12)):
        with contextlib.redirect_stdout(io.StringIO()):
            pred = GPGPUDynamicPredictor(256, 128, 2.4e9, bw, verbose=False, latency_mode="add", batch_size=256)
            r101_static = run(E, pred, "resnet101", "static", 1.0, None, None)
            rows = {
                "LAUD-ResNet101 channel-2222 keep 0.62": run(E, pred, "resnet101", "channel", 0.62, None, [2, 2, 2, 2]),
                "LAUD-ResNet101 spatial S=4-4-2-1 keep 0.5": run(E, pred, "resnet101", "spatial", 0.5, [4, 4, 2, 1], None),
                "LAUD-ResNet101 layer skip keep 0.5": run(E, pred, "resnet101", "layer", 0.5, None, None),
            }
            rg_static = run(E, pred, "regnety008", "static", 1.0, None, None)
            rg_layer = run(E, pred, "regnety008", "layer", 0.5, None, None)
        for name, lat in rows.items():
            out["rows"].append(dict(workload=name, mem_bandwidth=bw_name, static_latency_s=r101_static, dynamic_latency_s=lat,
                                    predicted_speedup=r101_static / lat))
        out["rows"].append(dict(workload="LAUD-RegNetY-800MF layer skip keep 0.5", mem_bandwidth=bw_name,
                                static_latency_s=rg_static, dynamic_latency_s=rg_layer, predicted_speedup=rg_static / rg_layer))
    path = os.path.join(ROOT, "profiles", "predicted_speedup_mi355x.json")
    json.dump(out, open(path, "w"), indent=1)
    for r in out["rows"]:
        print(f'{r["workload"]:48s} {r["mem_bandwidth"]:22s} static {r["static_latency_s"] * 1e3:8.3f} ms/img-batch  '
              f'dynamic {r["dynamic_latency_s"] * 1e3:8.3f}  predicted x{r["predicted_speedup"]:.2f}')


if __name__ == "__main__":
    main()
