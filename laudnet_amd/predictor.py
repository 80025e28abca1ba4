"""MI355X latency model of the HIP path ("MI355X-calibrated latency predictor ... with CU / LDS / Infinity-Cache
terms and MFMA instead of the scalar-lane model").

The reference predicts the latency of its dynamic operators analytically (DyNetSimulator/hardware_models/static_predictor.py
MultiCoresPredictor, multi_cores.py GPGPUDynamicPredictor: per-operator tile search over scalar fp32 lanes, an L2 / DRAM
bandwidth pair, latency = max or sum of a compute and a memory term, block formulas eval_example.py:12-122) and uses it to choose
granularities per hardware.  tools/predict_speedup.py drives that model unmodified (pinned to its V100 numbers).  THIS module is
the same idea re-derived for what the kernels of this repository actually do on CDNA4 -- it is not a restatement of the
reference's formulas:

  * work is counted in EXECUTED matrix-core FLOPs: three bf16 MFMA products per fp32 product (bf16x3), active channels padded to
    the 32-wide MFMA tile (E[32 ceil(K/32)] over the binomial number of active channel groups of an image), pixel tiles of 32;
  * a channel-mode block is one workgroup per (image, row block) on one CU: its time is a per-phase fixed cost + executed FLOPs /
    (the CU's MFMA rate x efficiency) + bytes through the CU's memory pipe / (per-CU stream rate) -- phases do not overlap inside a
    workgroup (measured), so the terms ADD (the reference's latency_mode "add");
  * bytes through the CU = activations (the block input twice: conv1 operand and residual; h1 out and back with the halo factor
    of the LDS-resident row block; the output) + the PER-IMAGE gathered weight subsets, which come from the XCD's L2 / the
    Infinity Cache, not from HBM;
  * a launch lasts ceil(workgroups / (CUs x workgroups per CU)) rounds (256 CUs; LDS decides the workgroups per CU);
  * density-independent launches (stem, projection shortcuts, the dense execution of stage 4, the stride-2 first blocks' gathered
    convs at their measured efficiency) are priced with a roofline: max(executed FLOPs / (2.5 PFLOP/s x eta_mfma), HBM bytes /
    (8 TB/s x eta_hbm)) + launch.

The free constants (per-CU MFMA efficiency, HBM efficiency of the activation streams, per-CU L2 gather rate, dense-kernel MFMA
efficiency, a per-forward residual) are FITTED to
measurements of this repository on MI355X (tools/calibrate_predictor.py -> profiles/r02_predictor_calibration.json: the step
time and the chained stage-3 block time of LAUD-ResNet101 bs256 at seven keep probabilities); tests/test_predictor.py checks
the fit against the committed measurements.  Use: `predict_resnet(...)`, `predicted_speedup(...)`, `best_channel_granularity(...)`.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import glob as _glob

_cals = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_predictor_calibration.json")))      # the latest round's fit (round 6: re-fitted on the
CALIBRATION = _cals[-1] if _cals else os.path.join(ROOT, "profiles", "r03_predictor_calibration.json")     # kernels of rounds 5-6: k_dense2 / k_rows3 / k_chain_ld)


@dataclass
class MI355X:
    """MI355X_MICROARCH.md: 256 CUs in 8 XCDs, 4 SIMDs per CU, v_mfma_f32_32x32x16_bf16 = 32768 FLOP in 32 cycles per SIMD
    (1024 FLOP / clk / SIMD -> 2.5 PFLOP/s dense bf16 at 2.4 GHz), 8 TB/s HBM3E, 256 MB Infinity Cache, 160 KB LDS per CU."""
    cus: int = 256
    simds: int = 4
    mfma_flop_per_clk_simd: float = 1024.0
    clock_hz: float = 2.4e9
    hbm_bytes_per_s: float = 8.0e12
    infinity_cache_bytes: float = 256e6
    lds_bytes: int = 160 * 1024

    @property
    def mfma_peak(self) -> float:
        return self.cus * self.simds * self.mfma_flop_per_clk_simd * self.clock_hz

    @property
    def cu_mfma_peak(self) -> float:
        return self.simds * self.mfma_flop_per_clk_simd * self.clock_hz


@dataclass
class Calibration:
    """Fitted constants (defaults = the round-2 fit; profiles/r02_predictor_calibration.json overrides them)."""
    cu_mfma_eff: float = 0.5          # fraction of a CU's MFMA peak a per-image workgroup sustains inside its matrix phases
    act_hbm_eff: float = 0.7          # fraction of 8 TB/s the per-image workgroups' activation streams reach together
    cu_l2_bytes_per_s: float = 60e9   # bytes / s one CU gathers its image's weight subsets from the XCD's L2 (LDS-DMA)
    phase_cost_s: float = 4.0e-6      # fixed cost per phase of a per-image workgroup (set-up 9 k + conversion 14 k cycles per block)
    dense_mfma_eff: float = 0.28      # shared-weight row kernels (k_dense / dense k_conv_bf3 / streaming 1x1): fraction of the chip's peak
    hbm_eff: float = 0.6              # fraction of 8 TB/s a streaming kernel reaches
    co_resident_gain: float = 1.3     # throughput gain of two co-resident workgroups per CU (64-wide layers) over one
    launch_s: float = 6.0e-6          # launch gap between dependent kernels
    fixed_s: float = 0.0              # residual per forward (bookkeeping kernels, pooling, classifier)
    # -- packed-row workloads (spatial / layer / LAD-RegNet layer skip; round 3)
    rows_cu_eff: float = 0.55         # k_dense: fraction of a CU's bf16 MFMA peak inside the K loop of a 256-row tile (executed FLOPs)
    narrow_alpha: float = 0.5         # ... x (tile columns / 256) ** narrow_alpha on narrower tiles (fewer MFMAs per staged activation row)
    tile_fixed_s: float = 8.0e-6      # prologue + epilogue of a tile's workgroup (tables, first chunk's round trip, transposes and stores)
    rows_hbm_eff: float = 0.5         # fraction of 8 TB/s the row kernels' gathers / scatters reach together
    idx_s: float = 30e-6              # index-list build per spatial block (two launches over the mask), beyond its bytes
    grouped_eff: float = 0.5          # k_grouped16_img: fraction of 8 TB/s on its rows in + rows out
    rows_fixed_s: float = 0.3e-3      # per-forward residual of the packed-row workloads (stem aside)
    # -- the one-launch small-map bottleneck (k_smallmap, stage 4; measured on the kernel alone: tools/bench_small.py at keep 0.5 / 0.62 / 0.75)
    small_step_s: float = 0.93e-6     # one K16 step of a workgroup (barrier + DMA burst + fragment reads; its 7-12 MFMAs hide under it)
    small_single_slot: float = 1.55   # ... x this when the image's channel list leaves a single ring slot (more than 480 of 512 channels kept)
    source: str = "defaults"

    @staticmethod
    def load(path: str = CALIBRATION) -> "Calibration":
        c = Calibration()
        if os.path.exists(path):
            d = json.load(open(path))
            for k, v in d.get("constants", {}).items():
                if hasattr(c, k):
                    setattr(c, k, v)
            c.source = os.path.relpath(path, ROOT)
        return c


# ------------------------------------------------------------------------------------------------ counting
def expected_padded_channels(groups: int, gran: int, density: float, tile: int = 32) -> float:
    """E[tile * ceil(K / tile)] with K = gran * Binomial(groups, density): the MFMA tile padding of an image's active subset."""
    if density >= 1.0:
        return float(tile * math.ceil(groups * gran / tile))
    if density <= 0.0:
        return 0.0
    e = 0.0
    logp, logq = math.log(density), math.log1p(-density)
    for k in range(groups + 1):
        lp = math.lgamma(groups + 1) - math.lgamma(k + 1) - math.lgamma(groups - k + 1) + k * logp + (groups - k) * logq
        e += math.exp(lp) * tile * math.ceil(k * gran / tile)
    return e


def expected_padded_sq(groups: int, gran: int, density: float, tile: int = 32) -> float:
    """E[(tile * ceil(K / tile))^2]: the 3x3 conv gathers inputs AND outputs."""
    if density >= 1.0:
        return float((tile * math.ceil(groups * gran / tile)) ** 2)
    if density <= 0.0:
        return 0.0
    e = 0.0
    logp, logq = math.log(density), math.log1p(-density)
    for k in range(groups + 1):
        lp = math.lgamma(groups + 1) - math.lgamma(k + 1) - math.lgamma(groups - k + 1) + k * logp + (groups - k) * logq
        e += math.exp(lp) * (tile * math.ceil(k * gran / tile)) ** 2
    return e


@dataclass
class BlockShape:
    cin: int
    width: int
    cout: int
    h_in: int
    w_in: int
    stride: int
    downsample: bool
    gran: int = 2

    @property
    def h(self):
        return self.h_in // self.stride

    @property
    def w(self):
        return self.w_in // self.stride


def resnet_blocks(layers=(3, 4, 23, 3), input_hw=(224, 224), gran=(2, 2, 2, 2), base=64):
    """Block shapes of a LAUD-ResNet (laud_resnet.py:208-250): [(stage, BlockShape)]."""
    h, w = input_hw[0] // 4, input_hw[1] // 4
    cin = base
    out = []
    for s, n in enumerate(layers):
        width = base * 2 ** s
        for j in range(n):
            stride = 2 if (j == 0 and s > 0) else 1
            out.append((s, BlockShape(cin, width, 4 * width, h, w, stride, j == 0, gran[s])))
            h, w = h // stride, w // stride
            cin = 4 * width
    return out


# ------------------------------------------------------------------------------------------------ kernel models
@dataclass
class Predictor:
    hw: MI355X = field(default_factory=MI355X)
    cal: Calibration = field(default_factory=Calibration.load)

    # -- per-image workgroups (k_head / k_tail / k_chain): one workgroup = one image x one block of output rows
    def rows_per_workgroup(self, b: BlockShape) -> tuple:
        """(output rows per workgroup, workgroups per image, halo factor of h1): at most 256 pixels per workgroup (8 waves x 32)."""
        r = max(1, min(b.h, 256 // b.w))
        mb = math.ceil(b.h / r)
        r = math.ceil(b.h / mb)
        return r, mb, min(r + 2, b.h) / r

    def fused_block(self, b: BlockShape, batch: int, density: float, chained: bool) -> dict:
        """conv1 + conv2 + conv3 of a stride-1 channel-mode block as per-image workgroups (two launches, or a share of the chained
        stage launch).  Returns seconds and the terms."""
        G = b.width // b.gran
        kp = expected_padded_channels(G, b.gran, density)
        kp2 = expected_padded_sq(G, b.gran, density)
        k64 = expected_padded_channels(G, b.gran, density, 64)
        r, mb, halo = self.rows_per_workgroup(b)
        px = r * b.w
        px_pad = 32 * math.ceil(px / 32)                       # one wave per 32 pixels
        flops = 6.0 * px_pad * (b.cin * kp + 9.0 * kp2 + kp * b.cout)          # executed: 3 bf16 products per fp32 product
        act = 4.0 * px * (b.cin + 2.0 * b.cout) + 4.0 * px * kp * (1.0 + halo)    # x, residual, out; h1 written once, staged with its halo
        wts = 4.0 * (k64 * b.cin + 9.0 * kp2 + kp * b.cout)                        # per-image gathered subsets (L2 / Infinity Cache)
        phases = 3 if chained else 4                             # masker | conv1 | conv2 | conv3 (chained: the masker rides along)
        # activations stream from HBM / the Infinity Cache while every CU does the same: a CU's share of the chip's rate;
        # the gathered weights are re-read by every image from the XCD's L2 at the per-CU rate
        act_rate = self.hw.hbm_bytes_per_s * self.cal.act_hbm_eff / self.hw.cus
        t_wg = phases * self.cal.phase_cost_s + flops / (self.hw.cu_mfma_peak * self.cal.cu_mfma_eff) + \
            act / act_rate + wts / self.cal.cu_l2_bytes_per_s
        wgs = batch * mb
        # 64-wide layers fit two workgroups per CU (LDS), whose matrix and memory phases overlap each other: priced as one
        # workgroup slot per CU running at 1 / co_resident_gain of the solo time
        gain = self.cal.co_resident_gain if b.width <= 64 else 1.0
        rounds = math.ceil(wgs / self.hw.cus)
        t = rounds * t_wg / gain
        if not chained:
            t += 2 * self.cal.launch_s
        return {"s": t, "flops": flops * wgs, "cu_bytes": (act + wts) * wgs, "rounds": rounds, "k_padded": kp}

    # -- shared-weight row kernels / density-independent launches: roofline with fitted efficiencies
    def dense_rows(self, rows: float, k: float, n: float, in_bytes: float, out_bytes: float) -> float:
        flops = 6.0 * rows * k * n
        t_m = flops / (self.hw.mfma_peak * self.cal.dense_mfma_eff)
        t_b = (in_bytes + out_bytes + 4.0 * k * n) / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff)
        return max(t_m, t_b) + self.cal.launch_s

    def strided_rows_per_workgroup(self, b: BlockShape) -> tuple:
        """(output rows per workgroup, workgroups per image, input pixels staged per K slice) of the fused tail on a stride-2 block:
        the region is staged one parity plane at a time -- the largest plane ((R + 1) x Wo plane pixels) must fit each of the TWO LDS
        slice slots beside the three W2 slots (csrc/ldn_tail.hip: tail_rows_per_block)."""
        ns = b.width // 32
        r = max(1, min(b.h, 256 // b.w))

        def fits(rr):
            plane = min(rr + 1, (b.h_in + 1) // 2) * b.w
            slice_b = -(-((-(-plane // 8) * 8 + 1) * 128) // 1024) * 1024
            return 1280 + 2 * slice_b + 3 * 16 * ns * 256 <= self.hw.lds_bytes
        while r > 1 and not fits(r):
            r -= 1
        mb = math.ceil(b.h / r)
        r = math.ceil(b.h / mb)
        return r, mb, min(b.stride * (r - 1) + 3, b.h_in) * b.w_in

    def first_block(self, b: BlockShape, batch: int, density: float) -> float:
        """The stage's first block (projection shortcut; stride 2 from stage 2 on) as executed since round 3: the dense projection
        (k_dense), conv1 on per-image workgroups at the INPUT resolution (k_head), conv2 (stride 2) -> conv3 + residual on per-image
        workgroups whose halo'd input region is four times their output block (k_tail<NS, 2>)."""
        G = b.width // b.gran
        kp = expected_padded_channels(G, b.gran, density)
        kp2 = expected_padded_sq(G, b.gran, density)
        k64 = expected_padded_channels(G, b.gran, density, 64)
        px_in, px = b.h_in * b.w_in, b.h * b.w
        act_rate = self.hw.hbm_bytes_per_s * self.cal.act_hbm_eff / self.hw.cus
        cu = self.hw.cu_mfma_peak * self.cal.cu_mfma_eff
        t = self.dense_rows(batch * px, b.cin, b.cout, 4.0 * batch * px * b.cin, 4.0 * batch * px * b.cout)             # projection
        # conv1: workgroups of <= 256 input pixels
        mb1 = math.ceil(px_in / 256)
        p1 = px_in / mb1
        t_wg = self.cal.phase_cost_s + 6.0 * 32 * math.ceil(p1 / 32) * b.cin * kp / cu + 4.0 * p1 * (b.cin + kp) / act_rate + \
            4.0 * k64 * b.cin / self.cal.cu_l2_bytes_per_s
        gain = self.cal.co_resident_gain if b.width <= 64 else 1.0
        t += math.ceil(batch * mb1 / self.hw.cus) * t_wg / gain + self.cal.launch_s
        # conv2 + conv3
        if b.stride == 1:
            r, mb, halo = self.rows_per_workgroup(b)
            region = r * b.w * halo
        else:
            r, mb, region = self.strided_rows_per_workgroup(b)
        p2 = r * b.w
        flops = 6.0 * 32 * math.ceil(p2 / 32) * (9.0 * kp2 + kp * b.cout)
        act = 4.0 * (region * kp + 2.0 * p2 * b.cout)
        wts = 4.0 * (9.0 * kp2 + kp * b.cout)
        t_wg = 2 * self.cal.phase_cost_s + flops / cu + act / act_rate + wts / self.cal.cu_l2_bytes_per_s
        g2 = self.cal.co_resident_gain if (b.width <= 64 and b.stride == 1) else 1.0
        t += math.ceil(batch * mb / self.hw.cus) * t_wg / g2 + self.cal.launch_s
        return t

    def dense_block(self, b: BlockShape, batch: int) -> float:
        """A channel-mode block executed densely (maps of <= 64 pixels: stage 4): masks applied in the epilogues, no work skipped."""
        px = b.h * b.w
        t = self.dense_rows(batch * b.h_in * b.w_in, b.cin, b.width, 4.0 * batch * b.h_in * b.w_in * b.cin, 4.0 * batch * b.h_in * b.w_in * b.width)
        t += self.dense_rows(batch * px, 9.0 * b.width, b.width, 4.0 * batch * b.h_in * b.w_in * b.width, 4.0 * batch * px * b.width)
        t += self.dense_rows(batch * px, b.width, b.cout, 4.0 * batch * px * (b.width + b.cout), 4.0 * batch * px * b.cout)
        if b.downsample:
            t += self.dense_rows(batch * px, b.cin, b.cout, 4.0 * batch * px * b.cin, 4.0 * batch * px * b.cout)
        return t + 2 * self.cal.launch_s                          # GAP + masker MLP launches

    def smallmap_block(self, b: BlockShape, batch: int, density: float) -> float:
        """A stride-1 identity block on a map of <= 64 pixels as ONE launch, one workgroup per image (ldn_bottleneck_smallmap): the
        workgroup walks cin / 16 K16 steps of conv1, 9 Kp / 16 of the 3x3 and (cout / 512) Kp / 16 of conv3 (Kp = the image's list padded
        to 32); a step costs the same whatever its MFMA count at these sizes (instruction issue and the per-step barrier bound it)."""
        G = b.width // b.gran
        kp = expected_padded_channels(G, b.gran, density)
        steps = b.cin / 16.0 + (9.0 + math.ceil(b.cout / 512.0)) * kp / 16.0
        t_wg = steps * self.cal.small_step_s * (self.cal.small_single_slot if kp > 480 else 1.0) + 2 * self.cal.phase_cost_s
        return math.ceil(batch / self.hw.cus) * t_wg + 2 * self.cal.launch_s        # + the masker's MLP launch

    @staticmethod
    def smallmap_fits(b: BlockShape) -> bool:
        """laud_resnet.Bottleneck._smallmap_eligible / ldn_bottleneck_smallmap_fits for the ResNet shapes."""
        return (b.stride == 1 and not b.downsample and b.cin == b.cout and b.h * b.w <= 64 and b.width <= 512 and b.width % 64 == 0
                and 2432 + (b.width // 32) * b.h * b.w * 128 + 32768 <= 160 * 1024 and b.gran % 2 == 0)

    def stem(self, batch: int, input_hw=(224, 224)) -> float:
        """k_stem: conv 7x7 (K padded to 176) on 17x15-pixel tiles recomputed 1.14x + pooled output."""
        px = batch * (input_hw[0] // 2) * (input_hw[1] // 2)
        flops = 6.0 * px * 1.14 * 176 * 64
        t_m = flops / (self.hw.mfma_peak * self.cal.dense_mfma_eff)
        t_b = 4.0 * batch * (3 * input_hw[0] * input_hw[1] + 64 * (input_hw[0] // 4) * (input_hw[1] // 4)) / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff)
        return max(t_m, t_b) + self.cal.launch_s

    # -- whole model
    def predict_resnet(self, batch: int = 256, layers=(3, 4, 23, 3), density=(0.62,) * 4, gran=(2, 2, 2, 2), input_hw=(224, 224)) -> dict:
        """Channel-mode LAUD-ResNet forward.  density: keep probability of a channel group per stage."""
        blocks = resnet_blocks(layers, input_hw, gran)
        rows, total = [], self.stem(batch, input_hw)
        rows.append(("stem", total))
        i = 0
        while i < len(blocks):
            s, b = blocks[i]
            d = density[s]
            if b.h * b.w <= 64 and self.smallmap_fits(b):                        # one launch per block (stage 4, blocks 2 and 3)
                t = self.smallmap_block(b, batch, d)
                kind = "smallmap"
            elif b.h * b.w <= 64:                                                # dense execution (stage 4's first block at 224)
                t = self.dense_block(b, batch)
                kind = "dense"
            elif b.downsample or b.stride != 1:
                t = self.first_block(b, batch, d)
                kind = "first"
            else:
                chained = b.h * b.w <= 256 and batch <= 4 * self.hw.cus
                t = self.fused_block(b, batch, d, chained)["s"]
                kind = "chained" if chained else "fused"
            rows.append((f"layer{s + 1}.{sum(1 for ss, _ in blocks[:i] if ss == s)} {kind}", t))
            total += t
            i += 1
        total += self.cal.fixed_s
        return {"s": total, "ms": 1e3 * total, "rows": rows}

    def predicted_speedup(self, batch=256, layers=(3, 4, 23, 3), density=(0.62,) * 4, gran=(2, 2, 2, 2), input_hw=(224, 224)) -> dict:
        """static (every channel kept: the same kernels at density 1) over dynamic -- eval_example.py:203-216 vs :219-360 for this
        implementation."""
        dyn = self.predict_resnet(batch, layers, density, gran, input_hw)
        sta = self.predict_resnet(batch, layers, (1.0,) * 4, gran, input_hw)
        return {"static_ms": sta["ms"], "dynamic_ms": dyn["ms"], "speedup": sta["ms"] / dyn["ms"]}

    # ================================================================================================ packed-row workloads (round 3)
    # spatial / layer modes of LAUD-ResNet and LAD-RegNet layer skip run on shared-weight row kernels over packed pixel lists
    # (k_dense: 256-row x NT-column tiles, one workgroup per CU; k_grouped16_img).  A launch is priced as
    #     t = max(rounds x t_tile, bytes / (8 TB/s x rows_hbm_eff)) + launch,     rounds = ceil(tiles / 256 CUs),
    #     t_tile = tile_fixed + executed FLOPs of a tile / (a CU's MFMA peak x rows_cu_eff x (NT / 256) ** narrow_alpha)
    # The ROUNDS are what the measurements show first: the layer workload's stage-3 3x3 (two 128-column tiles per 256 rows) takes
    # 160 us at keep 0.62 (244 workgroups: one round) and 261 us at keep 0.75 (294: two rounds), profiles/r03_density_sweep_layer.jsonl
    # and the kernel stats behind it (gpurun_out/r3h) -- +63 % time for +21 % rows.
    def tile_columns(self, n: int, m_cap: float, taps: int = 1) -> int:
        """The library's choice of the tile width (csrc/ldn_dense.hip: ldn_conv_rows_split)."""
        mt = math.ceil(m_cap / 256.0)
        if taps == 9:
            return 128 if (n % 128 == 0 or n > 64) else 64
        if n % 256 == 0 and mt * (n // 256) >= 384:
            return 256
        if n % 128 == 0:
            return 128
        if n <= 64:
            return 64
        if n % 160 == 0 or (n % 32 != 0 and n > 128):
            return 160
        return 128

    def _rows(self, rows: float, k: float, n: int, in_bytes: float, out_bytes: float, m_cap: float, taps: int = 1) -> float:
        if rows <= 0:
            return self.cal.launch_s
        nt = self.tile_columns(n, m_cap, taps)
        tiles = math.ceil(rows / 256.0) * math.ceil(n / nt)
        rounds = math.ceil(tiles / self.hw.cus)
        eff = self.cal.rows_cu_eff * min(1.0, nt / 256.0) ** self.cal.narrow_alpha
        t_tile = self.cal.tile_fixed_s + 6.0 * 256 * taps * k * nt / (self.hw.cu_mfma_peak * eff)
        t_b = (in_bytes + out_bytes + 4.0 * taps * k * n) / (self.hw.hbm_bytes_per_s * self.cal.rows_hbm_eff)
        return max(rounds * t_tile, t_b) + self.cal.launch_s

    def spatial_block(self, b: BlockShape, batch: int, s3: float, s1: float, layer_mode: bool = False, prev_s3: float = -1.0) -> dict:
        """One spatial- or layer-mode bottleneck: masker, index lists, conv1 on the dilated list (N1 rows), 3x3 on the output list
        (N3 rows, nine gathered h1 rows each), conv3 + residual scatter; first blocks add the dense projection.
        s3 / s1 = kept fraction of output / conv1 positions (the module's own sparsity outputs)."""
        px_in, px = b.h_in * b.w_in, b.h * b.w
        n1, n3 = s1 * batch * px_in, s3 * batch * px
        cap1, cap3 = batch * px_in, batch * px
        t = {}
        # masker: one pass over x -- of which only the images (layer mode) / patches (spatial mode) the previous block touched are
        # re-read when that block left the rest unchanged (stride 1, no projection: prev_s3 < 0 says it did not), the rest is carried
        frac = prev_s3 if (prev_s3 >= 0.0 and not b.downsample and b.stride == 1) else 1.0
        t["masker"] = 4.0 * frac * batch * px_in * b.cin / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff) + 2 * self.cal.launch_s
        t["index"] = (self.cal.launch_s if layer_mode else self.cal.idx_s + 2 * self.cal.launch_s) + 4.0 * 11 * n3 / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff)
        t["conv1"] = self._rows(n1, b.cin, b.width, 4.0 * n1 * b.cin, 4.0 * n1 * b.width, cap1)
        t["conv2"] = self._rows(n3, b.width, b.width, 4.0 * n1 * b.width, 4.0 * n3 * b.width, cap3, taps=9)
        t["conv3"] = self._rows(n3, b.width, b.cout, 4.0 * n3 * (b.width + b.cout), 4.0 * n3 * b.cout, cap3)
        if b.downsample:
            t["projection"] = self._rows(batch * px, b.cin, b.cout, 4.0 * batch * px * b.cin, 4.0 * batch * px * b.cout, cap3)
        t["s"] = sum(t.values())
        return t

    def predict_rows_resnet(self, batch: int, s3, s1, layers=(3, 4, 23, 3), input_hw=(224, 224), layer_mode: bool = False) -> dict:
        """Spatial- / layer-mode LAUD-ResNet forward from the per-block densities (lists over the blocks in execution order, as the
        module returns them / bench.py logs them under `block_densities`)."""
        blocks = resnet_blocks(layers, input_hw)
        total = self.stem(batch, input_hw) + self.cal.rows_fixed_s
        rows = [("stem", total)]
        prev = -1.0
        for i, (s, b) in enumerate(blocks):
            r = self.spatial_block(b, batch, s3[i], s1[i], layer_mode, prev)
            rows.append((f"layer{s + 1}.{i}", r["s"]))
            total += r["s"]
            prev = s3[i] if (not b.downsample and b.stride == 1) else -1.0      # a carry only from a block that leaves untouched units unchanged
        return {"s": total, "ms": 1e3 * total, "rows": rows}

    def predict_regnet_layerskip(self, batch: int, keep, widths=(64, 144, 320, 784), depths=(1, 3, 8, 2), stem_width=32, input_hw=(224, 224),
                                 group_width=16) -> dict:
        """LAD-RegNet-Y layer skip (BASELINE config 4): per block a (1x1) on the kept images' rows, b (grouped 3x3, HBM-bound),
        SE (three small launches over the kept rows), c (1x1) + residual; every stage's first block has stride 2 and a dense projection.
        keep = per-block kept fraction of images (list) or one number."""
        n_blocks = sum(depths)
        keep = list(keep) if hasattr(keep, "__len__") else [float(keep)] * n_blocks
        h, w = input_hw[0] // 2, input_hw[1] // 2
        px_stem = batch * h * w
        total = max(6.0 * px_stem * 48 * stem_width / (self.hw.mfma_peak * 0.1),
                    4.0 * batch * (3 * input_hw[0] * input_hw[1] + stem_width * h * w) / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff)) + self.cal.launch_s
        total += self.cal.rows_fixed_s
        rows = [("stem", total)]
        cin, i = stem_width, 0
        for s, (wd, d) in enumerate(zip(widths, depths)):
            for j in range(d):
                stride = 2 if j == 0 else 1
                ho, wo = h // stride, w // stride
                p = keep[i]
                n1, n3 = p * batch * h * w, p * batch * ho * wo
                mem = self.hw.hbm_bytes_per_s * self.cal.rows_hbm_eff
                t = 4.0 * batch * h * w * cin / (self.hw.hbm_bytes_per_s * self.cal.hbm_eff) * (1.0 if j == 0 else keep[i - 1]) + 3 * self.cal.launch_s   # masker + index
                t += self._rows(n1, cin, wd, 4.0 * n1 * cin, 4.0 * n1 * wd, batch * h * w)                              # a
                t += 4.0 * (n1 + n3) * wd / (self.hw.hbm_bytes_per_s * self.cal.grouped_eff) + self.cal.launch_s       # b
                t += 2.0 * 4.0 * n3 * wd / mem + 3 * self.cal.launch_s                                                 # SE: pool pass + scale pass
                t += self._rows(n3, wd, wd, 4.0 * n3 * 2.0 * wd, 4.0 * n3 * wd, batch * ho * wo)                       # c + residual
                if j == 0:
                    t += self._rows(batch * ho * wo, cin, wd, 4.0 * batch * ho * wo * cin, 4.0 * batch * ho * wo * wd, batch * ho * wo)   # projection
                rows.append((f"block{s + 1}-{j}", t))
                total += t
                cin, h, w = wd, ho, wo
                i += 1
        return {"s": total, "ms": 1e3 * total, "rows": rows}

    def best_channel_granularity(self, stage_shape: BlockShape, batch: int, density: float, candidates=(2, 4, 8, 16, 32)) -> dict:
        """Latency of a fused block per channel granularity at equal density: coarser groups waste less MFMA tile padding
        (E[32 ceil(K/32)] -> K as the group approaches the tile) -- the per-hardware choice the reference makes with its simulator."""
        out = {}
        for g in candidates:
            if stage_shape.width % g:
                continue
            b = BlockShape(**{**stage_shape.__dict__, "gran": g})
            out[g] = self.fused_block(b, batch, density, stage_shape.h * stage_shape.w <= 256)["s"]
        return out
