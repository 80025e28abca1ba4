#!/usr/bin/env python3
"""bench.py -- images/sec of the LAUDNet dynamic-inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one eval forward of the headline model over one synthetic batch that is already resident in HBM:
LAUD-ResNet101, channel granularity 2-2-2-2, MLP(2) maskers, target FLOPs ratio 0.5, 224x224, batch 256 per GPU
(BASELINE.json configs[1]).  Everything inside Bottleneck.forward runs in libldn_hip.so (maskers included --
the masks are produced by the maskers inside the timed region, nothing is injected or cached); the static stem,
average pool and classifier are library ops.  Weights are seeded random (no checkpoints offline), BN statistics
randomised; the channel maskers' keep-bias is calibrated once, before timing, so that the realised FLOPs ratio is
the "target-0.5" operating point of the released model.

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline     : the kernel with the most time per step -- algorithmic FLOPs / bytes per launch / mean launch duration measured with
                 HIP events on the launch stream in a separate leg of event-bracketed forwards right AFTER the timed region (the
                 timed region itself carries no instrumentation); `bound` is the larger of the HBM floor and the matrix floor
  cpu_baseline : the oracle (dense-emulation restatement of the reference, torch CPU) timed on the host cores on
                 a bounded sample of the same workload
and, unless --no-dense, `dense_emulation_gpu`: the same oracle run on the same GPU through PyTorch-ROCm
(the "reference dense-emulation PyTorch path" the north star's >=5x is quoted against).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# kernel arguments in device memory instead of host memory (a ROCm runtime switch, read when HIP initialises): every launch starts
# without a fetch over PCIe -- measured -3 % on the spatial workload, -6 % on RegNet, -0.5 % on the headline; it applies to
# every leg of this process alike (the oracle's dense emulation included).  HIP_FORCE_DEV_KERNARG=0 in the environment turns it off.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    "channel": dict(name="LAUD-ResNet101 channel-2222 target-0.5 @224",
                    kw=dict(dyn_mode=["channel"] * 4, channel_dyn_granularity=[2, 2, 2, 2], channel_masker=["MLP"] * 4,
                            channel_masker_layers=[2, 2, 2, 2], reduction_ratio=[16] * 4), p_channel=0.62, p_spatial=None),
    "spatial": dict(name="LAUD-ResNet101 spatial S=4-4-2-1 target-0.5 @224",
                    kw=dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[4, 4, 2, 1]), p_channel=None, p_spatial=0.5),
    "layer": dict(name="LAUD-ResNet101 layer-skip target-0.5 @224",
                  kw=dict(dyn_mode=["layer"] * 4), p_channel=None, p_spatial=0.5),
    # BASELINE config 1's model on the GPU: per-pixel masks -- every block runs the stand-alone spatial masker (one
    # pass over x), no pooled means to carry
    "spatial_g1": dict(name="LAUD-ResNet50 spatial g=1-1-1-1 (per-pixel masks) target-0.5 @224", arch="uni_resnet50", ref="resnet50_ref",
                       kw=dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[1, 1, 1, 1]), p_channel=None, p_spatial=0.5),
    # BASELINE config 4: the reference rejects dyn_mode='layer' for RegNet; layer skip = spatial with one patch per image
    "regnet": dict(name="LAUD-RegNetY-800MF layer-skip target-0.5 @224", arch="lad_regnet_y_800mf",
                   kw=dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[56, 28, 14, 7]), p_channel=None, p_spatial=0.5),
    # BASELINE config 5 (parity unpinned: the reference has no model code for it): DeiT-S shaped trunk of token-skipping blocks
    "adavit": dict(name="DeiT-S shaped token-skip trunk (12 blocks, 197 tokens, dim 384, 6 heads, MLP x4), token keep 0.5"),
}


def bench_adavit(args):
    """--workload adavit: the packed-token trunk of laudnet_amd/adavit.py (ldn_packed_mha + k_dense linears) on one GPU, beside its
    dense masked restatement (oracle/adavit_ref.py) run through PyTorch on the same GPU.  Keep masks: seeded Bernoulli per block,
    CLS always kept (the reference's token policy is not part of its repository).  One JSON line."""
    import laudnet_amd
    from laudnet_amd import ops
    from laudnet_amd.adavit import TokenSkipViT
    from oracle import adavit_ref as AR
    from fill import seeded_bernoulli, seeded_randn
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    laudnet_amd.load_library()
    ops.set_math_mode("bf16x3")
    depth, L, dim, heads = 12, 197, 384, 6
    keep_p = args.keep if args.keep is not None else 0.5
    ref = AR.TokenSkipViTRef(depth, dim, heads).eval()
    torch.manual_seed(1)
    for p_ in ref.parameters():
        if p_.dim() > 1:
            torch.nn.init.normal_(p_, std=0.03)
    hip = TokenSkipViT(depth, dim, heads).eval()
    hip.load_state_dict(ref.state_dict())
    hip, refg = hip.to(dev), ref.to(dev)
    x = seeded_randn((args.batch, L, dim), 1000).to(dev)
    keeps = []
    for i in range(depth):
        k = seeded_bernoulli((args.batch, L), keep_p, 2000 + i)
        k[:, 0] = 1.0
        keeps.append(k.to(dev))
    with torch.no_grad():
        for _ in range(args.warmup):
            out = hip(x, keeps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = hip(x, keeps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        for _ in range(3):
            want = refg(x, keeps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            want = refg(x, keeps)
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t0) / 10
    # roofline leg (after the timed region): HIP events around every shared-weight linear (k_dense) of two more forwards
    timer = KernelTimer()
    timer.PHASE = dict(timer.PHASE, rows_1x1=0)
    orig_rows = ops.conv_rows
    ops.conv_rows = timer.wrap_rows(orig_rows)
    try:
        with torch.no_grad():
            for i in range(2):
                timer.step = 2 * i       # every step sampled
                hip(x, keeps)
        agg = timer.summary()
    finally:
        ops.conv_rows = orig_rows
    roof = None
    if "rows_1x1" in agg:
        n_, ms_, f_, by_ = agg["rows_1x1"]
        roof = dict(two_floor("k_dense (the q/k/v, projection and MLP linears of the token-skip blocks over packed token rows, bf16x3; "
                              "averaged over all such launches of a forward)", n_, ms_, f_, by_, 3.0),
                    timed_ms_per_step=ms_ / 2, steps_bracketed=2)
    # head + layer skipping on top (simulate_adavit.py:81-88,140-182): per-image head masks (keep 0.7) and attention / MLP sub-block
    # decisions (keep 0.8), seeded; same trunk, same token masks
    extra = {}
    try:
        hks = [seeded_bernoulli((args.batch, heads), 0.7, 3000 + i).to(dev) for i in range(depth)]
        aks = [seeded_bernoulli((args.batch,), 0.8, 3100 + i).to(dev) for i in range(depth)]
        mks = [seeded_bernoulli((args.batch,), 0.8, 3200 + i).to(dev) for i in range(depth)]
        with torch.no_grad():
            for _ in range(2):
                o2 = hip(x, keeps, hks, aks, mks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(2, args.steps // 2)):
                o2 = hip(x, keeps, hks, aks, mks)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / max(2, args.steps // 2)
            w2 = refg(x, keeps, hks, aks, mks)
        extra["token_head_layer_skipping"] = {"value": args.batch / dt2, "unit": "images/sec", "ms_per_step": 1e3 * dt2,
                                              "head_keep": 0.7, "attention_block_keep": 0.8, "mlp_block_keep": 0.8,
                                              "max_abs_diff_vs_dense_restatement": (o2 - w2).abs().max().item()}
    except Exception as e:   # informative only
        extra["token_head_layer_skipping"] = {"error": repr(e)[:200]}
    kept = float(sum(k.sum().item() for k in keeps)) / (depth * args.batch * L)
    # algorithmic FLOPs per image.  EXECUTED: q / k / v, attention, projection and MLP all on the kept tokens (TokenSkipBlock.qkv_kept_only:
    # a dropped token is neither query nor key).  The reference's latency model prices q / k / v on EVERY token (simulate_adavit.py:90-93):
    # reported beside it as `reference_model_tflops`, not used for the roofline
    lk = kept * L
    lq = lk if TokenSkipViT.blocks_qkv_kept_only() else L
    flops = depth * 2.0 * (lq * dim * 3 * dim + 2 * lk * lk * dim + lk * dim * dim + 2 * lk * dim * 4 * dim)
    flops_ref_model = depth * 2.0 * (L * dim * 3 * dim + 2 * lk * lk * dim + lk * dim * dim + 2 * lk * dim * 4 * dim)
    result = {"metric": f"images/sec, DeiT-S shaped token-skip trunk @197 tokens bs{args.batch} (dynamic-token packed MHA)",
              "value": args.batch / dt, "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "bf16x3 (fp32 tensors; fp32 softmax / LayerNorm / GELU)", "data": "synthetic (seeded randn tokens, seeded random weights and keep masks)",
              "config": {"workload": WORKLOADS["adavit"]["name"].replace("keep 0.5", f"keep {keep_p}") + f" bs{args.batch}/GPU",
                         "kept_token_fraction": round(kept, 4), "algorithmic_tflops": flops * args.batch / dt / 1e12,
                         "reference_model_tflops": flops_ref_model * args.batch / dt / 1e12,
                         "parity": "UNPINNED: the reference holds no model code for this configuration (oracle/adavit_ref.py header)"},
              "roofline": roof,
              "dense_emulation_gpu": {"value": args.batch / dtd, "unit": "images/sec", "ms_per_step": 1e3 * dtd,
                                      "kind": "oracle/adavit_ref.py (dense masked attention, every token through every linear), PyTorch-ROCm fp32, same GPU",
                                      "max_abs_diff_vs_hip_same_masks": (out - want).abs().max().item(),
                                      "output_scale": want.abs().max().item()},
              "realised_speedup_vs_dense_emulation": dtd / dt, **extra}
    print(json.dumps(result))


def measure_chain_traffic(keep, batch):
    """HBM traffic of the dominant kernel (k_chain) measured ON THIS BOX: two separate rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE,
    WRITE_SIZE: the guide's recipe -- one TCC counter set per pass, no other trace domains) over a short product-path run of this script,
    read back from the rocpd database.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md).  Returns
    None when rocprofv3 is not usable here (the committed PMC pass of profiles/ is quoted then)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ldn_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--steps", "2",
                   "--warmup", "1", "--no-legs", "--batch", str(batch), "--keep", repr(float(keep))]
            subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None
            rows = sqlite3.connect(dbs[0]).execute("select dispatch_id, value from counters_collection where counter_name = ? and "
                                                   "kernel_name like '%k_chain%'", (ctr,)).fetchall()
            per = {}
            for disp, v in rows:
                per[disp] = per.get(disp, 0.0) + v
            if not per:
                return None
            vals[ctr] = min(per.values())     # KiB per dispatch; steady-state launches (the first also pulls the weights from HBM)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "fetch_kib_raw": vals["FETCH_SIZE"],
            "write_kib": vals["WRITE_SIZE"],
            "scope": "whole k_chain launch = 22 stage-3 blocks (9.14 GB algorithmic: every block's input read once + output written once); "
                     "FETCH_SIZE / WRITE_SIZE are counted at the L2-fabric boundary, so Infinity-Cache hits are included",
            "source": "MEASURED IN THIS RUN on this box: two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 2 "
                      "--warmup 1 --no-legs` right after the timed region; FETCH_SIZE doubled per MI355X_MICROARCH.md"}


KIND_KERNELS = {"rows_1x1": ("k_dense",), "rows_3x3": ("k_rows3",), "tail_fused": ("k_tail",), "grouped16_img": ("k_grouped16_img",),
                "conv2_3x3": ("k_conv_bf3", "k_conv_image"), "conv3_1x1": ("k_conv1x1_stream",)}


def measure_kind_traffic(workload, kind, keep, batch):
    """Mean HBM traffic PER LAUNCH of the kernels behind one roofline kind (KIND_KERNELS) on this box: the two PMC passes of measure_chain_traffic
    over `bench.py --workload W --no-legs`, averaged over every dispatch of those kernels in the profiled run (the roofline object of an
    aggregated kind -- "all shared-weight 1x1 launches of a step" -- averages its algorithmic bytes over the same launches)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    pats = KIND_KERNELS.get(kind)
    if not os.path.exists(exe) or not pats:
        return None
    vals, n_disp = {}, 0
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ldn_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--workload", workload,
                   "--steps", "2", "--warmup", "1", "--no-legs", "--batch", str(batch), "--keep", repr(float(keep))]
            subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", LDN_BENCH_NO_EVENTS="1"))
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None
            where = " or ".join("kernel_name like ?" for _ in pats)
            rows = sqlite3.connect(dbs[0]).execute(f"select dispatch_id, value from counters_collection where counter_name = ? and ({where})",
                                                   (ctr,) + tuple(f"%{q}%" for q in pats)).fetchall()
            per = {}
            for disp, v in rows:
                per[disp] = per.get(disp, 0.0) + v
            if not per:
                return None
            vals[ctr] = sum(per.values()) / len(per)     # KiB per dispatch, mean over the run's launches of these kernels
            n_disp = len(per)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "fetch_kib_raw": vals["FETCH_SIZE"],
            "write_kib": vals["WRITE_SIZE"], "dispatches_averaged": n_disp, "kernels": list(pats),
            "scope": f"mean over the {n_disp} launches of {' / '.join(pats)} in a profiled `bench.py --workload {workload} --steps 2 --warmup 1 --no-legs` "
                     "(calibration pass + three forwards); counted at the L2-fabric boundary (Infinity-Cache hits included)",
            "source": "two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE), FETCH_SIZE doubled per MI355X_MICROARCH.md"}


SECONDARY = ("spatial", "layer", "regnet", "adavit")     # BASELINE.json configs[2], the layer-skip ResNet, configs[3] (per-GPU shard), configs[4]


def run_secondary(args):
    """The other BASELINE configs, timed in the SAME invocation as the headline: each one is this script
    again (`--workload W --steps 30 --warmup 10 --brief`, its own process on the same GPU, after the headline's timed region and legs
    are done) -- 10 warm-up + 30 timed forwards at batch 256, masks produced by the maskers in the timed region, the oracle's dense
    emulation on the same GPU and the same-mask parity beside it.  Values are per-workload JSON lines reduced to the judged keys."""
    import subprocess
    out = {}
    for w in SECONDARY:
        # (30 timed forwards behind 10 warm-up ones: windows of ten forwards right after a process start read up to 2-3x slow now and then -- RegNet 10.8 vs 3.5 ms,
        # layer 14.1 vs 10.5 ms with every kernel at its usual duration in the event leg of the same process, i.e. a host-side stall of tens of milliseconds)
        n_steps, n_warm = "30", "10"
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", n_steps, "--warmup", n_warm, "--batch", str(args.batch), "--brief"]
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in pr.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            de = d.get("dense_emulation_gpu", {})
            roof = d.get("roofline") or {}
            out[w] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                      "steps": d["steps"], "warmup": d["warmup"], "dtype": d["dtype"].split(" ")[0],
                      "mean_block_flops_ratio": d["config"].get("mean_block_flops_ratio", d["config"].get("kept_token_fraction")),
                      "dense_emulation_gpu_ms_per_step": de.get("ms_per_step"),
                      "realised_speedup_vs_dense_emulation": d.get("realised_speedup_vs_dense_emulation"),
                      "max_abs_diff_vs_oracle_same_masks": de.get("max_abs_logit_diff_vs_hip_same_masks", de.get("max_abs_diff_vs_hip_same_masks")),
                      "output_scale": de.get("logit_scale", de.get("output_scale")),
                      "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches", "traffic", "traffic_over_algorithmic",
                                                                 "traffic_source", "algorithmic_mbytes_per_launch")} if roof else None,
                      "parity": d["config"].get("parity", "pinned (reference-generated fixtures, tests/golden)"),
                      "wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("same_kernels_keep_1.0", "alternating_batches"):
                if k in d:
                    out[w][k] = d[k]
            for k in ("roofline_rows_3x3", "mi355x_model", "predicted_speedup"):
                if k in d:
                    v = d[k]
                    out[w][k] = ({kk: v.get(kk) for kk in ("bound", "achieved", "peak", "unit", "frac", "frac_executed", "frac_mfma_executed", "avg_launch_us", "launches")}
                                 if k.startswith("roofline") else v)
        except Exception as e:   # a secondary workload must never take the headline line down
            out[w] = {"error": repr(e)[:300], "wall_s": round(time.perf_counter() - t0, 1)}
    return out


def two_floor(kernel, n, ms, flops, nbytes, mfma_mult):
    """HBM floor (algorithmic bytes at 8 TB/s) vs matrix floor (mfma_mult executed MFMA products per algorithmic product at
    the bf16 peak); `bound` = the larger floor, `frac` against it."""
    gbs = nbytes / (ms * 1e-3) / 1e9
    tf = flops / (ms * 1e-3) / 1e12
    hbm_us = nbytes / n / (HBM_PEAK_GBS * 1e9) * 1e6
    mfma_us = mfma_mult * flops / n / (BF16_MFMA_PEAK_TFLOPS * 1e12) * 1e6
    mb = mfma_us > hbm_us
    return {"kernel": kernel, "bound": "mfma" if mb else "hbm", "achieved": mfma_mult * tf if mb else gbs,
            "peak": BF16_MFMA_PEAK_TFLOPS if mb else HBM_PEAK_GBS, "unit": "TFLOP/s" if mb else "GB/s",
            "frac": (mfma_mult * tf / BF16_MFMA_PEAK_TFLOPS) if mb else gbs / HBM_PEAK_GBS,
            "floors_us_per_launch": {"hbm": hbm_us, "mfma_executed": mfma_us}, "frac_hbm": gbs / HBM_PEAK_GBS,
            "frac_mfma_executed": mfma_mult * tf / BF16_MFMA_PEAK_TFLOPS, "algorithmic_tflops": tf, "traffic": None,
            "launches": n, "avg_launch_us": 1e3 * ms / n, "algorithmic_mbytes_per_launch": nbytes / n / 1e6,
            "algorithmic_gflop_per_launch": flops / n / 1e9}


def blocks_of(model):
    """(module holding the maskers / forced masks, callable block) pairs in execution order"""
    if hasattr(model, "trunk_output"):          # LAD-RegNet: maskers live in block.f
        for blk in model.blocks():
            yield blk.f, blk
    else:
        for s in (1, 2, 3, 4):
            for blk in getattr(model, f"layer{s}"):
                yield blk, blk


def calibrate_maskers(model, x, p_channel, p_spatial):
    """One sequential pass: shift each masker's keep-logit bias so that the requested fraction of units is kept on
    this batch (a trained checkpoint would bring its own operating point; none is available offline)."""
    import torch.nn.functional as F
    with torch.no_grad():
        h = x.contiguous(memory_format=torch.channels_last)
        h = model.stem(h) if hasattr(model, "stem") else model.maxpool(model.relu(model.bn1(model.conv1(h))))
        state = (h, None, None, None, None, None, torch.tensor(0.0, device=x.device))
        for blk, runner in blocks_of(model):
            if blk.masker_channel is not None and p_channel is not None:
                mk = blk.masker_channel
                G = mk.channel_dyn_group
                _, _, _, logits = mk.lists(state[0], blk.channel_dyn_granularity, want_logits=True)
                diff = (logits[:, :G] - logits[:, G:]).flatten().float()
                shift = -torch.quantile(diff.cpu(), 1.0 - p_channel).item()
                last = mk.conv[-1] if mk.layers == 2 else mk.conv
                last.bias.data[:G] += shift
                mk._drop_cache()
            if blk.masker_spatial is not None and p_spatial is not None:
                ms = blk.masker_spatial
                g = ms.mask_channel_group
                _, _, _, logits = ms(state[0], 1.0, want_logits=True)
                diff = (logits[:, :g] - logits[:, g:]).flatten().float()
                shift = -torch.quantile(diff.cpu(), 1.0 - p_spatial).item()
                ms.conv.bias.data[:g] += shift
                ms._drop_cache()
            state = runner(state, 1.0)


def audit_masker_decisions(model, ref, x, ops, headline_math):
    """How many masker decisions of the HIP path differ from the ORACLE's own maskers (PyTorch fp32 on the same GPU) when
    both see the SAME block input, per arithmetic mode.  The block inputs are the HIP path's own, captured through the
    model's debug tap as CLONES with the in-place residual update left ON: the fused spatial / layer masker (decisions taken
    from the pooled means conv3's epilogue leaves, laud_resnet.py `_pool_grid`) therefore stays on the path that is audited,
    and every block is audited on the activations it really sees; a decision can only differ where |keep logit - drop logit|
    is within the two implementations' reduction-order noise.  `blocks_decided_from_pooled_means` counts the blocks whose
    decision came from the fused path."""
    res = {}
    rblocks = [(b.f if hasattr(b, "f") else b) for _, b in ref.blocks()]
    for math in ("fp32", "bf16x3"):
        ops.set_math_mode(math)
        flips = total = fused = 0
        worst_margin = 0.0
        taps = []
        model._tap = lambda j, blk, xin: taps.append((j, blk, xin.clone()))
        try:
            with torch.no_grad():
                model(x, 1.0)
                for j, blk, xin in taps:
                    fused += 1 if getattr(blk, "last_fused_decision", False) else 0
                    rb = rblocks[j]
                    if getattr(rb, "masker_channel", None) is not None and getattr(blk, "last_channel_mask", None) is not None:
                        lg = rb.masker_channel.logits(xin).reshape(xin.shape[0], 2, -1)
                        mine = blk.last_channel_mask.reshape(xin.shape[0], -1)
                    elif getattr(rb, "masker_spatial", None) is not None and getattr(blk, "last_spatial_mask", None) is not None:
                        lg = rb.masker_spatial.logits(xin).reshape(xin.shape[0], 2, -1)
                        mine = blk.last_spatial_mask.reshape(xin.shape[0], -1)
                    else:
                        continue
                    theirs = (lg[:, 0] >= lg[:, 1]).float()
                    diff = mine != theirs
                    flips += int(diff.sum())
                    total += diff.numel()
                    if bool(diff.any()):
                        worst_margin = max(worst_margin, float((lg[:, 0] - lg[:, 1]).abs()[diff].max()))
        finally:
            model._tap = None
        res[math] = {"decisions_differing_from_oracle_maskers": flips, "decisions_total": total,
                     "largest_oracle_logit_margin_at_a_differing_decision": worst_margin,
                     "blocks_decided_from_pooled_means": fused}
    ops.set_math_mode(headline_math)
    return res


def free_running_decisions(model, ref, x, ops, headline_math):
    """End-to-end decision drift: the ORACLE runs free (its own maskers on its own activations, nothing forced) and so does the HIP
    path; how many masker decisions differ?  (audit_masker_decisions is teacher-forced: same block inputs.)  A decision that flips
    early changes the activations of every later block, so this counts drift, not arithmetic error."""
    rblocks = [(b.f if hasattr(b, "f") else b) for _, b in ref.blocks()]
    recs, saved = {}, []
    for j, rb in enumerate(rblocks):
        for name in ("_channel_mask", "_spatial_mask"):
            if hasattr(rb, name):
                orig = getattr(rb, name)
                saved.append((rb, name, rb.forced_channel_mask, rb.forced_spatial_mask))

                def wrap(x_, t_, orig=orig, key=(j, name)):
                    out = orig(x_, t_)
                    recs[key] = out[0]
                    return out
                setattr(rb, name, wrap)
        rb.forced_channel_mask = rb.forced_spatial_mask = None
    res = {}
    try:
        with torch.no_grad():
            ref(x, 1.0)
            for math in ("fp32", "bf16x3"):
                ops.set_math_mode(math)
                model(x, 1.0)
                flips = total = nblk = 0
                first = None
                for j, (hb, _) in enumerate(blocks_of(model)):
                    for name, mine in (("_channel_mask", getattr(hb, "last_channel_mask", None)), ("_spatial_mask", getattr(hb, "last_spatial_mask", None))):
                        theirs = recs.get((j, name))
                        if mine is None or theirs is None:
                            continue
                        d = int((mine.reshape(mine.shape[0], -1) != theirs.reshape(theirs.shape[0], -1).to(mine.dtype)).sum())
                        flips += d
                        total += mine.numel()
                        if d:
                            nblk += 1
                            first = j if first is None else first
                res[math] = {"decisions_differing_from_free_running_oracle": flips, "decisions_total": total,
                             "blocks_with_a_difference": nblk, "first_block_with_a_difference": first}
    finally:
        for rb, name, fc, fs in saved:
            try:
                delattr(rb, name)       # drop the instance-level wrapper: the class method is back
            except AttributeError:
                pass
            rb.forced_channel_mask, rb.forced_spatial_mask = fc, fs
        ops.set_math_mode(headline_math)
    return res


class KernelTimer:
    """HIP-event timing of the channel-mode conv launches on the launch stream (torch's current stream == the stream
    the C ABI is given): two event records per launch.  Launches are classified as
      "conv2_3x3" : per-image channel-subset 3x3 (input and output lists)          -> MFMA roofline (FLOPs)
      "conv3_1x1" : 1x1 with gathered input channels, dense wide output + residual -> HBM roofline (bytes)."""

    def __init__(self):
        self.records = []
        self.rows = []
        self.tails = []
        self.chains = []
        self.grouped = []
        self.step = 0          # index of the timed step in flight (set by the timed loop)
        self.steps_of = {}     # kind -> set of steps in which its launches were bracketed

    # Two event records per launch cost ~4 us of stream time each (0.48 ms of a 19.7 ms step when every conv2 and conv3
    # launch is bracketed): each kind is bracketed in every OTHER timed step, which still samples the whole timed region.
    PHASE = {"conv2_3x3": 0, "conv3_1x1": 1, "rows_3x3": 0, "tail_fused": 1, "rows_1x1": 1, "grouped16_img": 0}
    EVERY_STEP = ("chain_fused",)   # one launch per forward: two event records per step cost nothing measurable

    def sampled(self, kind):
        if kind not in self.EVERY_STEP and (self.step + self.PHASE[kind]) % 2:
            return False
        self.steps_of.setdefault(kind, set()).add(self.step)
        return True

    def wrap(self, fn):
        def timed(*a, **kw):
            kind = None
            if kw.get("ksize", 1) == 3 and kw.get("k_cnt") is not None:
                kind = "conv2_3x3"
            elif kw.get("ksize", 1) == 1 and kw.get("k_cnt") is not None and kw.get("n_cnt") is None:
                kind = "conv3_1x1"
            if kind is None or not self.sampled(kind):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.records.append((kind, e0, e1, kw.get("k_cnt"), kw.get("n_cnt"), tuple(a[4].shape), tuple(a[1].shape),
                                 kw.get("residual") is not None))
            return out
        return timed

    def wrap_tail(self, fn):
        """ops.bottleneck_tail: the fused conv2 -> conv3 launch of a channel-mode block (k_tail)."""
        def timed(*a, **kw):
            if not self.sampled("tail_fused"):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.tails.append((e0, e1, a[4], tuple(a[0].shape), tuple(a[9].shape), kw.get("residual") is not None))
            return out
        return timed

    def wrap_chain(self, fn):
        """ops.bottleneck_chain: a run of stride-1 channel-mode blocks (stage 3) as ONE launch (k_chain)."""
        def timed(*a, **kw):
            self.sampled("chain_fused")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.chains.append((e0, e1, out[2], tuple(a[0].shape), a[3]))   # ch_cnt [n,B], x shape, width
            return out
        return timed

    def wrap_rows(self, fn):
        """ops.conv_rows (shared-weight packed-row convs of the spatial / layer / RegNet / token-skip paths, the projection
        shortcuts and the dense execution of stage 4): 3x3 launches as "rows_3x3", 1x1 launches as "rows_1x1"."""
        def timed(*a, **kw):
            kind = "rows_3x3" if kw.get("taps", 1) == 9 else "rows_1x1"
            if not self.sampled(kind):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            cap = kw.get("m_cap")
            if cap is None:
                cap = a[0].shape[0] if kw.get("a_rows") is None else kw["a_rows"].numel() // kw.get("taps", 1)
            self.rows.append((kind, e0, e1, kw.get("m_count"), cap, tuple(a[1].shape), kw.get("residual2d") is not None))
            return out
        return timed

    def wrap_rows_ps(self, fn):
        """ops.conv_rows_ps: the 1x1 launches of the pre-split packed path (conv1 writing pre-split h1, conv3 reading pre-split h2)."""
        def timed(*a, **kw):
            if not self.sampled("rows_1x1"):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            cap = kw.get("m_cap")
            if cap is None:
                cap = a[0].shape[0] if kw.get("a_rows") is None else kw["a_rows"].numel()
            self.rows.append(("rows_1x1", e0, e1, kw.get("m_count"), cap, tuple(a[1].shape), kw.get("residual2d") is not None))
            return out
        return timed

    def wrap_rows3(self, fn):
        """ops.conv3x3_rows_ps: the packed 3x3 on pre-split rows (k_rows3) -- "rows_3x3"."""
        def timed(*a, **kw):
            if not self.sampled("rows_3x3"):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            cap = kw.get("m_cap")
            if cap is None:
                cap = a[1].numel() // 9
            self.rows.append(("rows_3x3", e0, e1, kw.get("m_count"), cap, tuple(a[2].shape), False))
            self.rows3_kernel = True
            return out
        return timed

    def wrap_grouped_images(self, fn):
        """ops.grouped16_conv3x3_images: LAD-RegNet conv b (grouped 3x3, 16 channels per group) on whole kept images (k_grouped16_img)."""
        def timed(*a, **kw):
            if not self.sampled("grouped16_img"):
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.grouped.append((e0, e1, kw.get("m_count"), tuple(a[0].shape), tuple(a[4].shape), kw.get("images")))
            return out
        return timed

    def summary(self):
        """-> {kind: (launches, total ms, algorithmic FLOPs, algorithmic bytes)}.  Bytes: activations read once
        (4 * Ho*Wo * sum_b K_b), the weight matrix once, residual read and output written once."""
        torch.cuda.synchronize()
        agg = {}
        for kind, e0, e1, kc, nc, oshape, wshape, has_res in self.records:
            B, Ho, Wo, _ = oshape
            taps, cin, cout = wshape           # k-major weights [taps][cin][cout] (input channels gathered)
            kb = kc.double().cpu()
            nb = nc.double().cpu() if nc is not None else torch.full((B,), float(cout), dtype=torch.float64)
            flops = float((2.0 * Ho * Wo * taps * kb * nb).sum())
            nbytes = 4.0 * (Ho * Wo * float(kb.sum()) + taps * cin * cout + Ho * Wo * float(nb.sum()) * (2 if has_res else 1))
            n, ms, f, by = agg.get(kind, (0, 0.0, 0.0, 0.0))
            agg[kind] = (n + 1, ms + e0.elapsed_time(e1), f + flops, by + nbytes)
        for e0, e1, kc, hshape, oshape, has_res in self.tails:
            # algorithmic work of the fused tail per launch: conv2 on the active subsets + conv3; bytes = h1 read once (left-packed,
            # 4 B per element), residual read + output written once, both weight tensors once
            B, H, Wd, width = hshape
            cout = oshape[-1]
            kb = kc.double().cpu()
            flops = float((2.0 * H * Wd * (9.0 * kb * kb + kb * cout)).sum())
            nbytes = 4.0 * (H * Wd * float(kb.sum()) + 9 * width * width + width * cout + B * H * Wd * cout * (2 if has_res else 1))
            n, ms, f, by = agg.get("tail_fused", (0, 0.0, 0.0, 0.0))
            agg["tail_fused"] = (n + 1, ms + e0.elapsed_time(e1), f + flops, by + nbytes)
        for e0, e1, cnt, xshape, width in self.chains:
            # algorithmic work of a chained run per launch, block by block: conv1 + conv2 + conv3 on the image's active subset;
            # bytes with full fusion (4 * H^2 * (Cin + Cout)): the block's input read ONCE (conv1 operand and residual
            # are the same tensor), its output written once, the three weight tensors once.  h1 / h2 / masks are not counted.
            B, H, Wd, C = xshape
            kb = cnt.double().cpu()                       # [n, B]
            n_blk = kb.shape[0]
            flops = float((2.0 * H * Wd * (2.0 * C * kb + 9.0 * kb * kb)).sum())
            nbytes = n_blk * 4.0 * (2.0 * B * H * Wd * C + 2 * C * width + 9 * width * width)
            n, ms, f, by = agg.get("chain_fused", (0, 0.0, 0.0, 0.0))
            agg["chain_fused"] = (n + 1, ms + e0.elapsed_time(e1), f + flops, by + nbytes)
            self.chain_blocks = n_blk
        for kind, e0, e1, mc, cap, wshape, has_res in self.rows:          # w [cout][taps][cin]; rows = active output rows of the batch
            cout, taps, cin = wshape
            m = float(mc.item()) if mc is not None else float(cap)
            n, ms, f, by = agg.get(kind, (0, 0.0, 0.0, 0.0))
            agg[kind] = (n + 1, ms + e0.elapsed_time(e1), f + 2.0 * m * taps * cin * cout,
                         by + 4.0 * (m * (cin + cout * (2 if has_res else 1)) + taps * cin * cout))
        for e0, e1, mc, ashape, oshape, images in self.grouped:
            # grouped 3x3, 16 channels per group: 2 * 9 * 16 FLOPs per output element; every input row read once, every output row written once
            m_out = float(mc.item()) if mc is not None else float(oshape[0])
            C = oshape[1]
            rows_in = m_out * (ashape[0] / max(oshape[0], 1))
            n, ms, f, by = agg.get("grouped16_img", (0, 0.0, 0.0, 0.0))
            agg["grouped16_img"] = (n + 1, ms + e0.elapsed_time(e1), f + 2.0 * m_out * 9 * 16 * C, by + 4.0 * (rows_in + m_out) * C + 4.0 * 9 * 16 * C)
        return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="channel")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-emulation GPU baseline")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-legs", action="store_true", help="product path only: no fp32 leg, no dense emulation, no CPU baseline (profiling)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--brief", action="store_true", help="product path + roofline leg + dense emulation with same-mask parity only (no fp32 / "
                    "keep-1.0 / hipGraph legs, no decision audits, no CPU baseline): what the headline run uses for its `secondary` workloads")
    ap.add_argument("--no-pmc", action="store_true", help="headline run: quote the committed PMC pass of profiles/ for `roofline.traffic` instead of "
                    "measuring it in this run (two rocprofv3 --pmc passes, ~1.5 min)")
    ap.add_argument("--pmc-legs", action="store_true", help="measure `roofline.traffic` of the dominant kernel kind of a non-headline workload live "
                    "(two rocprofv3 --pmc passes, ~1.5 min); default: the committed pass of profiles/rNN_traffic.json is quoted")
    ap.add_argument("--no-secondary", action="store_true", help="headline run: do not append the `secondary` dict (BASELINE configs 3-5 "
                    "timed in the same invocation)")
    ap.add_argument("--graph", action="store_true", help="time the forward replayed as one hipGraph (same kernels, no launch "
                    "gaps; per-launch HIP events, hence the roofline, need eager launches -- the default run reports the graph "
                    "replay as the extra leg `hipgraph_replay`)")
    ap.add_argument("--keep", type=float, default=None, help="keep probability the maskers are calibrated to (default: the "
                    "workload's target-0.5 operating point: 0.62 for channel units, 0.5 for spatial / layer units)")
    ap.add_argument("--target-flops", type=float, default=None, help="calibrate the keep probability by bisection until the module's mean "
                    "FLOPs ratio (laud_resnet.py:146) reaches this value; default 0.5 = the 'target-0.5' of the workloads (the reference "
                    "trains towards that FLOPs ratio: utils/sparsity_loss_unify.py); --keep P fixes the keep probability instead "
                    "(SURVEY 8d's p = 0.62 / 0.5 realise 0.513 / 0.59)")
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default="bf16x3",
                    help="arithmetic of the MFMA convolutions (include/ldn_hip.h: ldn_set_math_mode); fp32 storage either way")
    args = ap.parse_args()
    if args.no_legs:
        args.no_dense = args.no_cpu = True
    if args.brief:
        args.no_cpu = True
    if args.workload == "adavit":
        return bench_adavit(args)

    import laudnet_amd
    from laudnet_amd import distributed as D
    from laudnet_amd import ops
    from fill import fill_state_dict, seeded_randn

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
        # 127.0.0.1) and hand over; rank 0 of that job prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    numa = D.pin_to_gpu_numa_node(local) if world > 1 else None     # host threads next to their GPU (launch latency x N processes)
    laudnet_amd.load_library()   # fail loudly if the HIP extension is missing
    ops.set_math_mode(args.math)

    wl = WORKLOADS[args.workload]
    kw = dict(wl["kw"], num_classes=1000, input_size=224)
    model = getattr(laudnet_amd, wl.get("arch", "uni_resnet101"))(**kw).eval()
    sd = fill_state_dict(model.state_dict(), 1)
    for k in sd:   # damp the residual branches (as zero_init_residual would) so 33 seeded-random blocks keep O(1) activations
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.3
    model.load_state_dict(sd)
    model = model.to(dev)
    torch.backends.cudnn.benchmark = True
    x = seeded_randn((args.batch, 3, 224, 224), 1000 + rank).to(dev).contiguous(memory_format=torch.channels_last)
    if args.keep is not None:
        wl = dict(wl, p_channel=args.keep if wl["p_channel"] is not None else None,
                  p_spatial=args.keep if wl["p_spatial"] is not None else None, name=wl["name"] + f" (keep {args.keep})")
    calibrate_maskers(model, x, wl["p_channel"], wl["p_spatial"])
    keep_used = wl["p_channel"] if wl["p_channel"] is not None else wl["p_spatial"]
    tf = args.target_flops if args.target_flops is not None else (0.5 if args.keep is None else None)
    if tf is not None:
        # "target-0.5" means the module-reported mean FLOPs ratio (laud_resnet.py:146), not the keep probability: a kept patch drags the
        # dilated conv1 region along (mask1), so keep 0.5 realises 0.59 in spatial mode; channel keep 0.62 realises 0.513.  Bisect the
        # keep probability the maskers are calibrated to until the ratio is 0.500 +- 0.002.
        lo, hi = 0.05, 1.0
        for _ in range(12):
            mid = 0.5 * (lo + hi)
            calibrate_maskers(model, x, mid if wl["p_channel"] is not None else None, mid if wl["p_spatial"] is not None else None)
            with torch.no_grad():
                r = model(x, 1.0)[5].float().mean().item()
            keep_used = mid
            if abs(r - tf) < 0.002:
                break
            lo, hi = (mid, hi) if r < tf else (lo, mid)
        wl = dict(wl, name=wl["name"] + f" (keep {keep_used:.3f} -> FLOPs ratio {r:.3f})")
    if world > 1:
        # replicas are replicas: every rank calibrated its maskers on its OWN shard (seed 1000 + rank) -- rank 0's calibrated weights are
        # broadcast so that all ranks run the same model (the reference loads one checkpoint everywhere, train/main.py:187)
        D.broadcast_state(model, src=0)
    calibrated_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ranks_seen = D.ranks_seen(dev) if world > 1 else 1       # all_reduce of ones over RCCL: the collective really spans N ranks

    graphed = None
    if args.graph:
        from laudnet_amd.laud_resnet import GraphedForward
        graphed = GraphedForward(model, x, 1.0)

    recompute = D.recompute_for(model, x.shape)

    def forward_local():
        with torch.no_grad():
            return graphed(x) if graphed is not None else model(x, 1.0)

    def step():
        return D.gather_outputs(forward_local(), recompute)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()

    # ---- the timed region: exactly args.steps forwards, no HIP events, no instrumentation of any kind.  The interpreter's cyclic garbage
    # collector is parked for its duration: a rare host-side stall of tens of milliseconds was seen twice in ~60 runs of the round (a 10-forward
    # window reading 14 ms per step instead of 10 with every kernel at its usual duration in the same process); a generation-2 pass over the
    # model's object graph is one candidate and costs nothing to rule out (20 alternating runs with / without: 11.03-11.09 ms either way)
    import gc
    gc.collect()
    if not os.environ.get("LDN_BENCH_KEEP_GC"):      # (A/B switch)
        gc.freeze()
        gc.disable()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for i in range(args.steps):
        # the exchange of batch i (all-gather of logits + all-reduce of sparsities, issued asynchronously on RCCL's stream)
        # overlaps the forward of batch i+1; every batch's global 7-tuple is completed inside the timed region
        nxt = D.gather_outputs_async(forward_local(), recompute)
        if pending is not None:
            out = pending.wait()
        pending = nxt
    out = pending.wait()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the roofline leg: the same forwards again, now with two HIP-event records (on the launch stream) around the launches
    # of the kernel kinds below; run AFTER the timed region so that the records cost nothing inside it
    timer = KernelTimer()
    orig_conv_image = ops.conv_image
    orig_conv_rows = ops.conv_rows
    orig_tail = ops.bottleneck_tail
    orig_chain = ops.bottleneck_chain
    orig_grouped = ops.grouped16_conv3x3_images
    orig_rows_ps, orig_rows3 = ops.conv_rows_ps, ops.conv3x3_rows_ps
    ev_steps = 0
    if rank == 0 and graphed is None and not os.environ.get("LDN_BENCH_NO_EVENTS"):
        ops.conv_image = timer.wrap(orig_conv_image)
        ops.conv_rows = timer.wrap_rows(orig_conv_rows)
        ops.bottleneck_tail = timer.wrap_tail(orig_tail)
        ops.bottleneck_chain = timer.wrap_chain(orig_chain)
        ops.grouped16_conv3x3_images = timer.wrap_grouped_images(orig_grouped)
        ops.conv_rows_ps = timer.wrap_rows_ps(orig_rows_ps)
        ops.conv3x3_rows_ps = timer.wrap_rows3(orig_rows3)
        ev_steps = max(2, min(args.steps, 10))
        try:
            t1 = time.perf_counter()
            for i in range(ev_steps):
                timer.step = i
                forward_local()
            torch.cuda.synchronize()
            ev_ms = 1e3 * (time.perf_counter() - t1) / ev_steps
        finally:
            ops.conv_image = orig_conv_image
            ops.conv_rows = orig_conv_rows
            ops.bottleneck_tail = orig_tail
            ops.bottleneck_chain = orig_chain
            ops.grouped16_conv3x3_images = orig_grouped
            ops.conv_rows_ps, ops.conv3x3_rows_ps = orig_rows_ps, orig_rows3
    if world > 1:
        torch.distributed.barrier()

    images = world * args.batch * args.steps
    # per-block density log (what the latency predictors take as input): s3, s2, s1 = kept fraction of conv3 / conv2 /
    # conv1 positions, cs = kept fraction of channels, per block in execution order (global-batch means)
    flat = lambda grp: [round(float(v), 4) for t in grp for v in t.reshape(-1).float().cpu()]
    block_densities = {"s3": flat(out[1]), "s2": flat(out[2]), "s1": flat(out[3]), "cs": flat(out[4]),
                       "flops_perc": [round(float(v), 4) for v in out[5].float().cpu()]}
    flops_perc = out[5].float().mean().item()
    flops_per_img = out[6].item()
    result = {
        "metric": "images/sec, " + wl["name"].split(" ")[0] + f" @224 bs{args.batch} (dynamic-inference hot path)",
        "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.math == "fp32" else "bf16x3 (fp32 tensors; each fp32 product = 3 bf16 MFMA products of hi/lo splits, fp32 accumulate)",
        "data": "synthetic (seeded randn images, seeded random weights, randomised BN stats)",
        "config": {"workload": wl["name"] + f" bs{args.batch}/GPU, masks produced by the maskers in the timed region",
                   "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                   "parallelism": f"dp{world} (batch shards, all-gather logits + all-reduce stats over RCCL)",
                   "mean_block_flops_ratio": round(flops_perc, 4), "module_macs_per_image": flops_per_img,
                   "keep_probability_calibrated_to": round(float(keep_used), 4),
                   "launch": "hipGraph replay" if args.graph else "eager",
                   "math_mode": args.math + (" (per-call argument of the C ABI; fp32 storage; the fp32-MFMA figure of the same "
                                             "workload is `fp32_mfma_mode`)" if args.math != "fp32" else ""),
                   "backend": (torch.distributed.get_backend() if world > 1 else "none (single process)"),
                   "world_size": (torch.distributed.get_world_size() if world > 1 else 1),
                   "rccl_ranks_seen": ranks_seen, "numa_node_rank0": numa,
                   "timed_region": "no instrumentation; the per-launch HIP events behind `roofline` are recorded in a separate "
                                   f"leg of {ev_steps} event-bracketed forwards right after it (`roofline.event_leg_ms_per_step`)"},
    }

    agg = timer.summary()
    if agg:
        # per-launch HBM traffic of the stage-3 instance of each kernel from the committed PMC passes (profiles/)
        import glob
        tjs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))   # the latest round's committed PMC pass
        tj = tjs[-1] if tjs else ""
        traffic = json.load(open(tj)) if (args.workload == "channel" and args.batch == 256 and os.path.exists(tj)) else {}
        if (rank == 0 and world == 1 and args.workload == "channel" and args.math == "bf16x3" and not args.no_legs and not args.brief
                and not args.no_pmc and "chain_fused" in agg):
            live = measure_chain_traffic(keep_used, args.batch)      # the dominant kernel's traffic on THIS box (falls back to the committed pass)
            if live is not None:
                traffic = dict(traffic, chain_fused_bf16x3=live)
        mode = args.math

        def mfma_roofline(n, ms, flops, nbytes):
            achieved = flops / (ms * 1e-3) / 1e12
            if mode == "fp32":
                kernel, peak, extra = "k_conv_image (3x3 per-image channel-subset conv, fp32 MFMA)", F32_MFMA_PEAK_TFLOPS, {}
            else:   # three bf16 MFMAs are executed per algorithmic product: the matrix pipe sees 3x the algorithmic FLOPs
                kernel, peak = "k_conv_bf3 (3x3 per-image channel-subset conv, bf16x3 split-precision MFMA)", BF16_MFMA_PEAK_TFLOPS
                extra = {"executed_mfma_tflops": 3 * achieved, "frac_executed": 3 * achieved / peak,
                         "frac_of_fp32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS}
            return {"kernel": kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": traffic.get(f"conv2_3x3_{mode}", {}).get("traffic_bytes_per_launch"),
                    "traffic_scope": "stage-3 launch (22.8 GFLOP algorithmic)", "launches": n, "avg_launch_us": 1e3 * ms / n,
                    "algorithmic_gflop_per_launch": flops / n / 1e9, **extra}

        def hbm_roofline(n, ms, flops, nbytes):
            achieved = nbytes / (ms * 1e-3) / 1e9
            return {"kernel": ("k_conv_image" if mode == "fp32" else "k_conv1x1_stream") + " (1x1 conv3: gathered input channels, "
                              "dense wide output, residual + ReLU epilogue)", "bound": "hbm", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic.get(f"conv3_1x1_{mode}", {}).get("traffic_bytes_per_launch"),
                    "traffic_scope": "stage-3 launch (443 MB algorithmic)", "launches": n, "avg_launch_us": 1e3 * ms / n,
                    "algorithmic_mbytes_per_launch": nbytes / n / 1e6, "algorithmic_tflops": flops / (ms * 1e-3) / 1e12}

        def tail_roofline(n, ms, flops, nbytes):
            # the fused tail is bounded by HBM (h1 + residual in, output out: 442 MB per stage-3 launch = 55 us at 8 TB/s) slightly
            # before the matrix pipe (3 bf16 MFMA products per algorithmic product: 118 GFLOP executed = 47 us at 2.5 PFLOP/s)
            achieved = nbytes / (ms * 1e-3) / 1e9
            tf = flops / (ms * 1e-3) / 1e12
            return {"kernel": "k_tail (fused conv2 3x3 -> bn2/ReLU -> conv3 1x1 -> bn3 + residual + ReLU of a channel-mode block, bf16x3)",
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic.get("tail_fused_bf16x3", {}).get("traffic_bytes_per_launch"),
                    "traffic_scope": traffic.get("tail_fused_bf16x3", {}).get("scope"),
                    "launches": n, "avg_launch_us": 1e3 * ms / n, "algorithmic_mbytes_per_launch": nbytes / n / 1e6,
                    "algorithmic_tflops": tf, "executed_mfma_tflops": 3 * tf, "frac_of_bf16_mfma_peak_executed": 3 * tf / BF16_MFMA_PEAK_TFLOPS}

        def chain_roofline(n, ms, flops, nbytes):
            # whole stage-3 run in one launch; with full fusion a block moves its input once and its output once (411 MB at bs256).
            # Two floors: HBM (algorithmic bytes at 8 TB/s) and the matrix pipe (3 bf16 MFMA products per algorithmic product at
            # 2.5 PFLOP/s); `bound` names the LARGER one (the binding roof), `frac` is measured against it, both fractions are given.
            achieved = nbytes / (ms * 1e-3) / 1e9
            tf = flops / (ms * 1e-3) / 1e12
            nb = getattr(timer, "chain_blocks", 1)
            hbm_floor_us = nbytes / n / (HBM_PEAK_GBS * 1e9) * 1e6
            mfma_floor_us = 3.0 * flops / n / (BF16_MFMA_PEAK_TFLOPS * 1e12) * 1e6
            mfma_binds = mfma_floor_us > hbm_floor_us
            tr = traffic.get("chain_fused_bf16x3", {})
            return {"kernel": "k_chain_ld (`ldn::k_chain_ld<8>` in the rocprof stats: the loader / consumer form of k_chain; one launch = a run of "
                              "stride-1 channel-mode bottlenecks: masker -> conv1 -> conv2 3x3 -> conv3 + residual per block, one workgroup per image, bf16x3)",
                    "bound": "mfma" if mfma_binds else "hbm",
                    "achieved": 3 * tf if mfma_binds else achieved, "peak": BF16_MFMA_PEAK_TFLOPS if mfma_binds else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if mfma_binds else "GB/s",
                    "frac": (3 * tf / BF16_MFMA_PEAK_TFLOPS) if mfma_binds else achieved / HBM_PEAK_GBS,
                    "floors_us_per_launch": {"hbm": hbm_floor_us, "mfma_bf16x3_executed": mfma_floor_us},
                    "frac_hbm": achieved / HBM_PEAK_GBS, "achieved_gbs_algorithmic": achieved,
                    "frac_mfma_executed": 3 * tf / BF16_MFMA_PEAK_TFLOPS, "executed_mfma_tflops": 3 * tf, "algorithmic_tflops": tf,
                    "traffic": tr.get("traffic_bytes_per_launch"), "traffic_scope": tr.get("scope"),
                    "traffic_source": tr.get("source", "committed PMC pass (profiles/), NOT measured in this run"),
                    "launches": n, "avg_launch_us": 1e3 * ms / n, "blocks_per_launch": nb, "avg_us_per_block": 1e3 * ms / n / nb,
                    "algorithmic_mbytes_per_launch": nbytes / n / 1e6, "algorithmic_mbytes_per_block": nbytes / n / nb / 1e6}

        def rows1_roofline(n, ms, flops, nbytes):
            return two_floor("k_dense2 / k_dense (shared-weight 1x1 over packed pixel rows: conv1 / conv3 / projection shortcuts / RegNet a, c / "
                             "token-skip linears; bf16x3, averaged over all such launches of a step)", n, ms, flops, nbytes, 3.0 if mode != "fp32" else 16.0)

        def grouped_roofline(n, ms, flops, nbytes):
            return two_floor("k_grouped16_img (LAD-RegNet conv b: grouped 3x3, 16 channels per group, whole kept images staged in LDS, bf16x3)",
                             n, ms, flops, nbytes, 3.0 * 10.0 / 9.0)   # five K steps of two taps for nine taps

        pick = {"conv3_1x1": hbm_roofline, "tail_fused": tail_roofline, "chain_fused": chain_roofline, "rows_1x1": rows1_roofline,
                "grouped16_img": grouped_roofline}
        objs = {k: pick.get(k, mfma_roofline)(*v) for k, v in agg.items()}
        if "rows_3x3" in objs:
            objs["rows_3x3"]["kernel"] = ("k_rows3 (3x3 conv over packed active rows on PRE-SPLIT h1 through the neighbour table, shared weights in K64 steps, bf16x3)"
                                          if getattr(timer, "rows3_kernel", False) else
                                          "k_dense2<.., T9> (3x3 conv over packed active rows through the neighbour table, shared weights, bf16x3; rows split between the MFMA steps)"
                                          if mode != "fp32" and 9 in ops.DENSE_TAPS and ops.USE_DENSE_KERNEL else
                                          objs["rows_3x3"]["kernel"].replace("3x3 per-image channel-subset conv", "3x3 conv over packed active rows, shared weights"))
            objs["rows_3x3"]["traffic_scope"] = objs["rows_3x3"]["traffic"] = None
        # dominant = most time per bracketed step inside the timed region
        per_step = lambda k: agg[k][1] / max(len(timer.steps_of.get(k, ())), 1)   # ms of kind k per bracketed step
        order = sorted(agg, key=lambda k: -per_step(k))
        result["roofline"] = dict(objs[order[0]], timed_ms_per_step=per_step(order[0]), steps_bracketed=len(timer.steps_of.get(order[0], ())),
                                  event_leg_ms_per_step=ev_ms)
        for k in order[1:]:
            result["roofline_" + k] = dict(objs[k], timed_ms_per_step=per_step(k), steps_bracketed=len(timer.steps_of.get(k, ())))
        if result["roofline"].get("traffic") is None and order[0] != "chain_fused":
            #  the wasted-traffic ratio of the dominant kind of the other workloads: measured here with --pmc-legs,
            # else quoted from the latest committed pass (profiles/rNN_traffic.json, key "<workload>:<kind>")
            import glob as _glob
            key = f"{args.workload}:{order[0]}"
            tr = measure_kind_traffic(args.workload, order[0], keep_used, args.batch) if (args.pmc_legs and world == 1) else None
            if tr is not None:
                tr["source"] = "MEASURED IN THIS RUN: " + tr["source"]
            else:
                tjs_ = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
                tr = (json.load(open(tjs_[-1])).get(key) if tjs_ and args.batch == 256 else None)
                if tr is not None:
                    tr = dict(tr, source="committed PMC pass (" + os.path.basename(tjs_[-1]) + "), NOT measured in this run: " + tr.get("source", ""))
            if tr is not None:
                result["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                result["roofline"]["traffic_scope"] = tr.get("scope")
                result["roofline"]["traffic_source"] = tr.get("source")
                alg = result["roofline"].get("algorithmic_mbytes_per_launch")
                if alg:
                    result["roofline"]["traffic_over_algorithmic"] = tr["traffic_bytes_per_launch"] / (alg * 1e6)
                result["roofline"]["traffic_detail"] = {k_: tr.get(k_) for k_ in ("fetch_kib_raw", "write_kib", "dispatches_averaged", "kernels")}

    result["block_densities"] = block_densities
    result.setdefault("roofline", None)   # workloads whose hot kernels are not timed per launch (RegNet grouped conv, --graph)
    if rank == 0 and world == 1 and not args.graph and not args.no_legs and not args.brief:
        # same forward, same kernels, replayed as one hipGraph (launch gaps removed)
        try:
            from laudnet_amd.laud_resnet import GraphedForward
            saved = [(hb, getattr(hb, "last_channel_mask", None), getattr(hb, "last_spatial_mask", None)) for hb, _ in blocks_of(model)]
            gf = GraphedForward(model, x, 1.0)
            with torch.no_grad():
                for _ in range(2):
                    outg = gf(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    outg = gf(x)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            result["hipgraph_replay"] = {"value": args.batch / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt,
                                         "max_abs_logit_diff_vs_eager": (outg[0] - out[0]).abs().max().item()}
            del gf
            for hb, cm, sm in saved:
                hb.last_channel_mask, hb.last_spatial_mask = cm, sm
        except Exception as e:   # informative only
            result["hipgraph_replay"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and args.math != "fp32" and not args.no_legs and not args.graph and not args.brief:
        # the same workload with fp32 operands on v_mfma_f32_32x32x2_f32 (reported beside the headline, not as `value`)
        def grab_masks():
            return [m for hb, _ in blocks_of(model) for m in (getattr(hb, "last_channel_mask", None), getattr(hb, "last_spatial_mask", None))
                    if m is not None]
        masks_head = grab_masks()
        ops.set_math_mode("fp32")

        for _ in range(2):
            out32 = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out32 = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        masks32 = grab_masks()
        total = sum(m.numel() for m in masks_head)
        flips = sum(int((a != b).sum().item()) for a, b in zip(masks_head, masks32))
        # masker decisions are thresholds: a 1e-6 perturbation flips the few that sit on a tie, after which the two runs
        # follow different masks -- the arithmetic parity figure is dense_emulation_gpu.max_abs_logit_diff_vs_hip_same_masks
        result["fp32_mfma_mode"] = {"value": args.batch / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt,
                                    "masker_decisions_differing_from_headline_mode": flips, "masker_decisions_total": total,
                                    "max_abs_logit_diff_vs_headline_mode_own_masks": (out[0] - out32[0]).abs().max().item()}
        ops.set_math_mode(args.math)
        out = step()   # leave the modules' last_*_mask in the headline mode for the same-mask parity leg below
        torch.cuda.synchronize()
    if (rank == 0 and world == 1 and not args.no_legs and not args.graph and wl["p_spatial"] is not None):
        # Two DIFFERENT batches alternating: the timed region replays one resident batch, so the row-count hints the
        # packed-row kernels size their tiles with (ops.RowsHint = the PREVIOUS forward's device-side counts) are exact there.  Here every forward
        # sees the counts of the other batch: the hint is stale by one forward, as on a stream of real batches.  Results are hint-independent
        # (tests); this leg measures what the staleness costs.
        try:
            x_other = seeded_randn((args.batch, 3, 224, 224), 2000 + rank).to(dev).contiguous(memory_format=torch.channels_last)
            pair = (x, x_other)
            with torch.no_grad():
                for i in range(4):
                    model(pair[i & 1], 1.0)
                torch.cuda.synchronize()
                n_alt = max(6, min(args.steps, 20)) & ~1
                t0 = time.perf_counter()
                for i in range(n_alt):
                    o_alt = model(pair[i & 1], 1.0)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n_alt
                ratios = [float(model(pair[i], 1.0)[5].float().mean().item()) for i in range(2)]
            result["alternating_batches"] = {"ms_per_step": 1e3 * dt, "value": args.batch / dt, "unit": "images/sec", "steps": n_alt,
                                             "mean_block_flops_ratio_of_the_two_batches": [round(r, 4) for r in ratios],
                                             "vs_resident_batch": 1e3 * dt / result["ms_per_step"],
                                             "note": "two seeded batches alternate: every forward's row-count hints are the other batch's counts"}
            del x_other, o_alt
            out = step()     # the resident batch's masks again (the parity leg below replays them)
            torch.cuda.synchronize()
        except Exception as e:   # informative only
            result["alternating_batches"] = {"error": repr(e)[:200]}
    if (rank == 0 and world == 1 and not args.no_legs and not args.graph
            and (wl["p_channel"] is not None or wl["p_spatial"] is not None)):
        # the SAME kernels with every unit kept (maskers recalibrated to keep 1.0): what the dynamic masks buy on this implementation
        # (in --brief runs too since round 6: the secondary legs report it as `mi355x_model.realised_speedup_of_the_masks`)
        try:
            calibrate_maskers(model, x, 1.0 if wl["p_channel"] is not None else None, 1.0 if wl["p_spatial"] is not None else None)
            for _ in range(2):
                out1 = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                out1 = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 4
            result["same_kernels_keep_1.0"] = {"value": args.batch / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt,
                                               "mean_block_flops_ratio": round(out1[5].float().mean().item(), 4),
                                               "realised_speedup_of_the_masks": 1e3 * dt / result["ms_per_step"]}
        except Exception as e:   # informative only
            result["same_kernels_keep_1.0"] = {"error": repr(e)[:200]}
        model.load_state_dict({k: v.to(dev) for k, v in calibrated_sd.items()})
        out = step()
        torch.cuda.synchronize()
    if rank == 0 and world == 1:
        from oracle import torch_ref as TR
        if "ref" in wl:
            ref = getattr(TR, wl["ref"])(**kw).eval()
        elif "arch" in wl:
            from oracle import regnet_ref as RR
            ref = RR.regnet_y_ref(wl["arch"], **kw).eval()
        else:
            ref = TR.resnet101_ref(**kw).eval()
        ref.load_state_dict(calibrated_sd)
        if not args.no_dense:
            try:
                refg = ref.to(dev).to(memory_format=torch.channels_last)
                # parity on identical inputs AND masks: replay the masks the HIP maskers produced in the last step
                for (hb, _), rb in zip(blocks_of(model), ((b.f if hasattr(b, "f") else b) for _, b in refg.blocks())):
                    cm, sm = getattr(hb, "last_channel_mask", None), getattr(hb, "last_spatial_mask", None)
                    rb.forced_channel_mask = None if cm is None else cm.clone()
                    rb.forced_spatial_mask = None if sm is None else sm.clone()
                with torch.no_grad():
                    for _ in range(10):            # MIOpen autotune on, 10 warm-up + 50 timed iterations
                        want = refg(x, 1.0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    reps = 50
                    for _ in range(reps):
                        want = refg(x, 1.0)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                err = (out[0] - want[0]).abs().max().item()
                result["dense_emulation_gpu"] = {"value": args.batch / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt,
                                                 "kind": "oracle dense emulation, PyTorch-ROCm fp32 channels_last, same GPU, "
                                                         "MIOpen autotune on, 10 warm-up + 50 timed forwards",
                                                 "max_abs_logit_diff_vs_hip_same_masks": err,
                                                 "logit_scale": want[0].abs().max().item()}
                result["realised_speedup_vs_dense_emulation"] = result["value"] / (args.batch / dt)
                # BASELINE.md publishes no number for this metric; the baseline the north star names (">= 5x the reference dense-emulation
                # PyTorch path's images/sec") is measured in THIS run on THIS GPU: vs_baseline = value / that
                result["vs_baseline"] = result["realised_speedup_vs_dense_emulation"]
                result["vs_baseline_note"] = ("value / dense_emulation_gpu.value, both measured in this run on this GPU (BASELINE.md 3: the >= 5x "
                                              "denominator); BASELINE.md holds no published number for this metric")
                if "fp32_mfma_mode" in result:   # like-for-like: fp32 multiply on both sides
                    result["realised_speedup_fp32_mode"] = result["fp32_mfma_mode"]["value"] / (args.batch / dt)
                if not args.brief:
                    try:
                        result["masker_decision_audit"] = audit_masker_decisions(model, refg, x, ops, args.math)
                    except Exception as e:   # informative only
                        result["masker_decision_audit"] = {"error": repr(e)[:200]}
                    try:
                        result["masker_decision_audit"]["free_running"] = free_running_decisions(model, refg, x, ops, args.math)
                    except Exception as e:   # informative only
                        result["masker_decision_audit"]["free_running"] = {"error": repr(e)[:200]}
                ref = ref.cpu()
            except Exception as e:  # the baseline is informative only
                result["dense_emulation_gpu"] = {"error": repr(e)[:200]}
        if not args.no_cpu:
            host_cores = os.cpu_count() or 1
            # thread count: measured, not assumed -- one forward of a small batch at 8 / 16 / 32 / 64 / 128 threads (torch CPU convs stop
            # scaling far below 256 threads on this host: 16 threads beat 32 by 1.8x on the EPYC 9575F box), the fastest count runs the
            # timed sample
            for rb in ((b.f if hasattr(b, "f") else b) for _, b in ref.blocks()):
                rb.forced_channel_mask = rb.forced_spatial_mask = None
            ref = ref.cpu()
            xs_ = x[: min(args.cpu_batch, 8)].cpu().contiguous()
            sweep = {}
            for n_thr in sorted({min(host_cores, c) for c in (8, 16, 32, 64, 128)}):
                torch.set_num_threads(n_thr)
                with torch.no_grad():
                    ref(xs_[:2], 1.0)
                    t0 = time.perf_counter()
                    ref(xs_, 1.0)
                    sweep[n_thr] = xs_.shape[0] / (time.perf_counter() - t0)
            cores = max(sweep, key=sweep.get)
            torch.set_num_threads(cores)
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                cpu_model = "unknown"
            for rb in ((b.f if hasattr(b, "f") else b) for _, b in ref.blocks()):
                rb.forced_channel_mask = rb.forced_spatial_mask = None
            xc = x[: args.cpu_batch].cpu().contiguous()
            ref = ref.cpu()
            with torch.no_grad():
                ref(xc[:2], 1.0)
                t0 = time.perf_counter()
                reps = 0
                while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 8):
                    ref(xc, 1.0)
                    reps += 1
                dt = (time.perf_counter() - t0) / reps
            result["cpu_baseline"] = {"value": args.cpu_batch / dt, "unit": "images/sec", "cores": cores, "kind": "port",
                                      "host": f"{host_cores} logical cores, {cpu_model}",
                                      "thread_sweep_images_per_sec": {str(k): round(v, 1) for k, v in sweep.items()},
                                      "sample": f"{reps} forward passes of batch {args.cpu_batch} (same model/weights/inputs), "
                                                f"oracle dense emulation in torch fp32 on {cores} threads"}
    # second half of the BASELINE metric: the reference's analytic predictor with MI355X parameters (tools/predict_speedup.py,
    # generated in the build container by importing the reference's DyNetSimulator; committed under profiles/)
    pj = os.path.join(ROOT, "profiles", "predicted_speedup_mi355x.json")
    if os.path.exists(pj):
        key = {"channel": "channel-2222", "spatial": "spatial S=4-4-2-1", "layer": "ResNet101 layer skip", "regnet": "RegNetY-800MF"}.get(args.workload)
        pd = json.load(open(pj))
        rows = [r for r in pd["rows"] if key is not None and key in r["workload"]]
        unc = [r for r in rows if r["mem_bandwidth"].startswith("8.0 TB/s (spec)")]
        cal = [r for r in rows if "calibrated" in r["mem_bandwidth"]]
        if unc:
            result["predicted_speedup"] = {"value": unc[0]["predicted_speedup"], "source": "reference DyNetSimulator (harness pinned to its V100 "
                                           "numbers by tests/test_predictor.py), MI355X parameters (256 CUs x 128 fp32 lanes, 2.4 GHz, "
                                           "8 TB/s), bs256; reference-default (NVIDIA-fitted) knobs"}
            if cal:
                result["predicted_speedup"]["calibrated"] = {
                    "value": cal[0]["predicted_speedup"], "knobs": pd.get("calibration", {}).get("knobs"),
                    "fitted_to": "9 dense fp32 convs measured on MI355X (profiles/r02_dense_convs.json)",
                    "predicted_static_ms_per_batch": 1e3 * cal[0]["static_latency_s"]}
            best = (cal or unc)[0]["predicted_speedup"]
            if "realised_speedup_vs_dense_emulation" in result:
                result["realised_over_predicted"] = {"math_" + args.math: result["realised_speedup_vs_dense_emulation"] / best}
                if "realised_speedup_fp32_mode" in result:   # the predictor models fp32 FMA lanes: the like-for-like comparison
                    result["realised_over_predicted"]["math_fp32"] = result["realised_speedup_fp32_mode"] / best
                result["realised_over_predicted"]["predicted"] = best
                k1 = result.get("same_kernels_keep_1.0", {})
                if "realised_speedup_of_the_masks" in k1:   # static baseline = the same kernels with nothing skipped (not eager PyTorch)
                    result["realised_over_predicted"]["vs_same_kernels_keep_1.0"] = k1["realised_speedup_of_the_masks"] / best
    if rank == 0:
        # the MI355X-native latency model of THIS implementation (laudnet_amd/predictor.py; constants fitted on four keep probabilities
        # of profiles/r03_density_sweep_*.jsonl, validated on three held-out ones): predicted step time from the per-block densities of
        # this run, and its predicted speedup over the same kernels with every unit kept
        try:
            from laudnet_amd.predictor import Predictor
            P = Predictor()
            bd = block_densities
            if args.workload == "channel":
                dyn = P.predict_resnet(args.batch, density=(float(keep_used),) * 4)["ms"]
                sta = P.predict_resnet(args.batch, density=(1.0,) * 4)["ms"]
            elif args.workload == "regnet":
                dyn = P.predict_regnet_layerskip(args.batch, bd["s3"])["ms"]
                sta = P.predict_regnet_layerskip(args.batch, [1.0] * len(bd["s3"]))["ms"]
            else:
                lm = args.workload == "layer"
                lay = (3, 4, 6, 3) if WORKLOADS[args.workload].get("arch", "").endswith("resnet50") else (3, 4, 23, 3)
                dyn = P.predict_rows_resnet(args.batch, bd["s3"], bd["s1"], layers=lay, layer_mode=lm)["ms"]
                sta = P.predict_rows_resnet(args.batch, [1.0] * len(bd["s3"]), [1.0] * len(bd["s1"]), layers=lay, layer_mode=lm)["ms"]
            k1 = result.get("same_kernels_keep_1.0", {})
            result["mi355x_model"] = {
                "predicted_ms_per_step": dyn, "predicted_ms_with_everything_kept": sta, "predicted_speedup_of_the_masks": sta / dyn,
                "measured_over_predicted_ms": result["ms_per_step"] / dyn,
                "realised_speedup_of_the_masks": k1.get("realised_speedup_of_the_masks"),
                "note": "static baseline = the same HIP kernels with every unit kept (`same_kernels_keep_1.0`, measured in this run); the "
                        "constants were fitted on another box of the pool (boxes differ by up to ~9 %)",
                "calibration": P.cal.source}
            if args.workload == "spatial_g1":
                result["mi355x_model"]["note"] += "; NOT fitted on per-pixel masks (its masker term is the pooled-mean masker's): indicative only"
        except Exception as e:   # informative only
            result["mi355x_model"] = {"error": repr(e)[:200]}
    if (rank == 0 and world == 1 and args.workload == "channel" and not args.no_legs and not args.brief and not args.graph
            and not args.no_secondary and args.keep is None and args.target_flops is None and args.math == "bf16x3"):
        result["secondary"] = run_secondary(args)
    if rank == 0:
        try:
            result["plan_timeouts"] = ops.plan_timeouts()      # list-build launches that ran into their time bound (0 = healthy)
        except Exception as e:
            result["plan_timeouts"] = repr(e)[:100]
        # the judged headline figures once more, LAST on the line (a truncated tail of the line still carries them)
        de = result.get("dense_emulation_gpu", {})
        result["headline"] = {"ms_per_step": result["ms_per_step"], "images_per_sec": result["value"],
                              "dense_emulation_gpu_ms_per_step": de.get("ms_per_step"),
                              "realised_speedup_vs_dense_emulation": result.get("realised_speedup_vs_dense_emulation"),
                              "max_abs_logit_diff_vs_oracle_same_masks": de.get("max_abs_logit_diff_vs_hip_same_masks"),
                              "roofline_frac": (result.get("roofline") or {}).get("frac"),
                              "rccl_ranks_seen": result["config"].get("rccl_ranks_seen")}
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
