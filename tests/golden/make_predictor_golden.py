#!/usr/bin/env python3
"""Pins the predictor harness (tools/predict_speedup.py): runs the REFERENCE's own DyNetSimulator/eval_example.py main body,
unmodified, for resnet50 / resnet101 / regnety008 on its V100 preset (80 SMs x 64 lanes, 1.5 GHz, 700 GB/s, batch 128,
eval_example.py:137-156) and stores the four latencies it computes but never prints (static / spatial / layer / channel,
eval_example.py:203-360) in tests/golden/predictor_v100.json.  Build container only (needs /root/reference); numbers only."""
import contextlib
import io
import json
import os
import runpy

import numpy as np
import sys

REF = "/root/reference/DyNetSimulator"
out = {"source": "reference DyNetSimulator/eval_example.py run as __main__ (unmodified), --hardware v100", "models": {}}
os.chdir(REF)
sys.path.insert(0, REF)
for model in ("resnet50", "resnet101", "regnety008"):
    sys.argv = ["eval_example.py", model, "--hardware", "v100"]
    np.random.seed(0)   # the channel-mode latency draws random group masks (hardware_models/utils.py:32): seeded, same order of calls
    with contextlib.redirect_stdout(io.StringIO()):
        g = runpy.run_path(os.path.join(REF, "eval_example.py"), run_name="__main__")
    out["models"][model] = {k: float(g[k]) for k in ("static_latency", "s_latency", "l_latency", "c_latency")}
    print(model, out["models"][model])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "predictor_v100.json"), "w"), indent=1)
