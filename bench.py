#!/usr/bin/env python
"""bench.py — lip-clips/s, forward+backward, of the SBL hot path on synthetic 29x88x88 clips (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                          # one rank per GPU, RCCL over xGMI

A "step" = one optimizer-free training step of the reference (SBL/train.py:188-196): Transformer.forward on a
batch of 32 clips per GPU (Conv3d stem -> ResNet-18 -> 6-layer encoder -> 16-step SBL decoder), the two
label-smoothed losses, loss.backward(), and for N > 1 the gradient average over ranks.  Dropout is ON (it is part
of the reference step), BatchNorm in training mode; tensors are fp32 in memory throughout.  --precision picks the
arithmetic of the GEMM / convolution tile engine (include/sbl_hip.h, sbl_set_matmul_precision): the default
"bf16x6" is the fp32-grade exact three-way bf16 split with six bf16 MFMA products and fp32 accumulation (every GPU
parity test runs under it and under "f32" with the same tolerances); "f32" is the exact fp32 MFMA; "bf16" is BASELINE
config 5's mixed precision (--workload config5: T=64, 112x112, 16 clips per GPU).  Inputs are synthetic and already
resident in HBM.  The step is captured into hipGraphs after the warm-up and replayed: dropout masks change per replay (their
seed lives in device memory); the 16 teacher-forcing coins (decoder.py:176) determine which decoder steps can be
batched, hence the launch sequence, so --coin-patterns random patterns are drawn (seed 7, same on every rank), one
graph is captured per pattern and the timed loop cycles through them.

The coin patterns are a stratified sample of Binomial(16, 1/2) (draw_coin_patterns: own-arg-max counts 5..11, mean exactly 8 =
the expectation; round 2 used four raw draws with mean 7.25, i.e. ~0.45 ms less decoder work per step), the timed loop
visits each pattern equally often (24 steps over 8 patterns), and `config.per_pattern` reports every pattern's own time.

Prints ONE JSON line (rank 0).  Extra objects:
  f32_exact    — the same step re-captured and replayed under the exact fp32 MFMA (--precision f32 arithmetic), same run, same
                 patterns: the like-for-like number next to the split-bf16 headline (N = 1 only)
  roofline     — the dominant kernel family of the step by GPU time.  Every GEMM / convolution launch of the step
                 is timed live, inside replays of the same captured step, by in-kernel s_memrealtime stamps
                 (min start / max end over the launch's workgroups; include/sbl_hip.h sbl_profile_*), i.e. on the
                 stream the kernel runs on and with the real inter-kernel concurrency.  achieved = algorithmic
                 FLOPs of the family's launches / their summed durations.  `families` lists the others.
  cpu_baseline — the CPU oracle (oracle/sbl_oracle.py, the fixture-pinned restatement of the reference) timed on the
                 host cores of this box on a bounded sample, N=1 / rank 0 only
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_FRAMES, HW = 29, 88
PER_GPU_BATCH = 32
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_TBS = 8.0
# fp32-equivalent MFMA peak of each precision mode: the bf16 modes spend 6 / 3 / 1 bf16 MFMA products per fp32 product
MODE_PEAK_TFLOPS = {"f32": FP32_MFMA_PEAK_TFLOPS, "bf16x6": BF16_MFMA_PEAK_TFLOPS / 6, "bf16x3": BF16_MFMA_PEAK_TFLOPS / 3,
                    "bf16": BF16_MFMA_PEAK_TFLOPS}
MODE_DTYPE = {"f32": "f32", "bf16x6": "f32 (exact 3-way bf16 split, 6 bf16 MFMA products per fp32 product, fp32 accumulate)",
              "bf16x3": "bf16x3 (2-way bf16 split, 3 products, fp32 accumulate)", "bf16": "bf16 (fp32 master weights and accumulation)"}
KERNEL_NAMES = {1: "sbl_skinny_gemm_kernel (decoder/encoder nn.Linear fwd/dX/dW, M<=512)",
                2: "sbl_mfma_gemm_kernel / sbl_mfma_gemm2_kernel 64x64 dense (nn.Linear, M>512; gemm2 = both decoder directions per launch)",
                3: "sbl_mfma_gemm_kernel 128x128 dense",
                4: "trunk conv fwd + BN stats: sbl_conv_patch_kernel (22x22, 11x11 maps), sbl_conv_pm_kernel (6x6, 3x3), sbl_mfma_gemm_kernel<ConvGatherKC,DenseKC> (stride 2, 1x1)",
                5: "trunk conv input grad: sbl_conv_patch_kernel<dgrad> (22x22, 11x11), sbl_conv_pm_kernel<dgrad> (6x6, 3x3), sbl_conv_classes_kernel (stride 2: parity classes, one launch)",
                6: "trunk conv weight grad: sbl_conv_patch_wgrad_kernel (3x3 stride 1 at 22x22, 11x11, 6x6), sbl_conv_pm_wgrad_kernel (3x3 maps), sbl_mfma_gemm_kernel<DenseMC,ConvGatherMC> (stride 2, 1x1; split-K atomics)",
                7: "sbl_wgrad_group_kernel<SegMC,SegMC 128x128> (all deferred decoder / encoder weight grads, one launch each, no split-K)",
                8: "stem (Conv3d 5x7x7 fwd + BN/ReLU/pool + backward reduce + weight gradient; the two contractions follow the matmul precision)",
                9: "encoder self-attention (29 frames: two query tiles per (batch, head) on the one-wavefront attention_small kernels; config 5's 64 frames: attention_fwd/bwd_kernel, one workgroup per (batch, head)); fp32 MFMA"}
# kernel-name patterns of each family in the rocprofv3 --pmc summary (profiles/*_pmc_fetch_write_per_kernel.csv)
KERNEL_PMC_RE = {1: r"sbl_skinny_gemm_kernel", 2: r"sbl_mfma_gemm2?_kernel<Dense[KM]C<64, \w+>, Dense[KM]C<64, \w+>, EpiStore",
                 3: r"sbl_mfma_gemm_kernel<Dense[KM]C<128, \w+>, Dense[KM]C<128, \w+>, EpiStore",
                 4: r"((sbl_mfma_gemm_kernel|sbl_conv_pm_kernel)<ConvGather(KC|PM)<\d+, false>|sbl_conv_patch_kernel<\d, false)",
                 5: r"((sbl_mfma_gemm_kernel|sbl_conv_pm_kernel|sbl_conv_classes_kernel)<ConvGather(KC|PM)<\d+, true>|sbl_conv_patch_kernel<\d, true)",
                 6: r"(sbl_mfma_gemm_kernel<DenseMC<\d+, true>, ConvGatherMC|sbl_conv_pm_wgrad_kernel|sbl_conv_patch_wgrad_kernel)", 7: r"sbl_wgrad_group_kernel"}
PMC_SUMMARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_fetch_write_per_kernel.csv")
T_START = time.perf_counter()


def pmc_traffic(kid):
    """HBM-side bytes per launch of a kernel family from the round's rocprofv3 PMC passes of this same command
    (tools/pmc_passes.sh regenerates the CSV: FETCH_SIZE x the factor calibrated on copies of known size,
    profiles/r02_pmc_calibration.txt, + WRITE_SIZE; counters cannot be collected from inside the process).
    None when the summary is missing."""
    import csv, re
    try:
        rows = list(csv.DictReader(open(PMC_SUMMARY)))
    except OSError:
        return None
    n = b = 0.0
    for r in rows:
        if re.search(KERNEL_PMC_RE.get(kid, "$^"), r["kernel"]):
            k = float(r["launches"])
            n += k
            b += k * 1024.0 * (float(r["avg_FETCH_SIZE_KB_calibrated"]) + float(r["avg_WRITE_SIZE_KB"]))
    return round(b / n) if n else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default 32; 16 for --workload config5)")
    ap.add_argument("--workload", default="full", choices=["full", "frontend", "config5"],
                    help="full = BASELINE config 3 (the metric's configuration); frontend = config 2; config5 = T=64, 112x112, mixed bf16")
    ap.add_argument("--precision", default=None, choices=("f32", "bf16x6", "bf16x3", "bf16"),
                    help="arithmetic of the GEMM / convolution tile engine (default bf16x6; bf16 for --workload config5)")
    ap.add_argument("--no-graph", action="store_true", help="run eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--mode", choices=("auto", "graph", "eager"), default="auto",
                    help="how the step is issued: replayed hipGraphs, eager launches, or (auto) whichever a short trial of both "
                         "finds faster on this box (same on every rank)")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the instrumented replays behind `roofline`")
    ap.add_argument("--cpu-batch", type=int, default=0, help="clips per CPU-oracle step (default 32 ~ 6 s on 16 cores; 4 for config5)")
    ap.add_argument("--coin-patterns", type=int, default=8, help="distinct teacher-forcing coin patterns (one graph each)")
    ap.add_argument("--no-pack-cache", action="store_true", help="A/B: re-pack the 19 convolution weights inside every step (round-2 behaviour)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: set an attribute of the decoder module (e.g. fuse_stage_io=0, fuse_kv_projections=0) or, with an "
                         "'ops.' prefix, of the ops module, before the step is built")
    ap.add_argument("--tuning", action="append", default=[], metavar="KNOB=VALUE", help="A/B: sbl_set_tuning(knob, value) before the step is built")
    ap.add_argument("--no-f32-exact", action="store_true", help="skip the secondary exact-fp32 measurement (`f32_exact` in the JSON line)")
    ap.add_argument("--per-step-decoder", action="store_true", help="one decoder stage per step (no run batching)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--single-stream", action="store_true", help="run the two decoder directions on one stream")
    ap.add_argument("--split-graph", choices=("auto", "on", "off"), default="auto",
                    help="capture the step as two hipGraphs (everything up to the encoder's input gradient | frontend backward) so "
                         "that the decoder + encoder gradient all-reduces overlap the frontend backward; auto = on when N > 1")
    ap.add_argument("--dump-launches", default="", help="debug: write a per-shape table of the instrumented launches to this file")
    ap.add_argument("--hang-dump", type=int, default=0, help="debug: dump all Python stacks after this many seconds")
    ap.add_argument("--verbose", action="store_true", help="progress lines on stderr")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = "bf16" if args.workload == "config5" else "bf16x6"
    if args.batch <= 0:
        args.batch = 16 if args.workload == "config5" else PER_GPU_BATCH
    args.T, args.HW = (64, 112) if args.workload == "config5" else (T_FRAMES, HW)
    if args.cpu_batch <= 0:
        args.cpu_batch = 4 if args.workload == "config5" else 32
    return args


def draw_coin_patterns(n, seed=7):
    """n teacher-forcing coin patterns (decoder.py:176: 16 fair coins per step) whose own-argmax counts are the mid-quantiles
    of Binomial(16, 1/2) - for n = 8: 5, 6, 7, 8, 8, 9, 10, 11, mean 8 = the expectation.  The step time is linear in the
    count (one more sequential decoder stage per own-argmax coin), so a stratified sample measures the expected step time
    without the luck of n raw draws (Random(7)'s first four have 4, 8, 8, 9: mean 7.25).  The patterns themselves are the
    first draws of Random(seed) with each wanted count."""
    from math import comb
    cdf, acc = [], 0.0
    for k in range(17):
        acc += comb(16, k) / 65536.0
        cdf.append(acc)
    want = [next(k for k in range(17) if cdf[k] >= (i + 0.5) / n) for i in range(n)]
    rng = random.Random(seed)
    out = [None] * n
    left = list(range(n))
    while left:
        p = [rng.random() > 0.5 for _ in range(16)]
        for i in left:
            if want[i] == sum(p):
                out[i] = p
                left.remove(i)
                break
    # Order: the timed steps use patterns 1, 2, ..., n - 1, 0, 1, ... (step 0 is untimed).  Middle-out, alternating below / above
    # the mean, keeps every PREFIX of that sequence near the expectation (8, 8, 7, 9, 6, 10, 5, 11: prefix means 8, 8, 7.7, 8,
    # 7.6, 8, 7.6, 8), so a run with --steps that is not a multiple of n is not flattered (sorted order: 5 steps -> mean 6.8).
    lo, hi = (n - 1) // 2, n // 2
    seq = []
    while lo >= 0 or hi < n:
        if lo == hi:
            seq.append(out[lo])
        else:
            if lo >= 0:
                seq.append(out[lo])
            if hi < n:
                seq.append(out[hi])
        lo, hi = lo - 1, hi + 1
    return [seq[-1]] + seq[:-1]


def log(args, msg):
    if args.verbose:
        print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def build_model(device, dropout_on):
    from sbl_for_multilingual_lip_reading_amd import detfill
    from sbl_for_multilingual_lip_reading_amd.transformer.decoder import Decoder
    from sbl_for_multilingual_lip_reading_amd.transformer.encoder import Encoder
    from sbl_for_multilingual_lip_reading_amd.transformer.transformer import Transformer
    # SBL/utils.py:90-114 defaults; SBL/train.py:58-69
    enc = Encoder(512, 6, 8, 64, 64, 512, 2048, dropout=0.1, pe_maxlen=5000)
    dec = Decoder(0, 1, 58, 512, 6, 8, 64, 64, 512, 2048, dropout=0.1, tgt_emb_prj_weight_sharing=True, pe_maxlen=5000)
    m = Transformer(enc, dec, None)
    sd = m.state_dict()
    m.load_state_dict({k: (v if k.endswith(".pe") else torch.from_numpy(detfill.fill_value(k, tuple(v.shape)).copy()))
                       for k, v in sd.items()})
    if not dropout_on:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.visual_frontend.frontend_dropout_p = 0.0
    return m.to(device).train()


class LaunchRecorder:
    """While active, notes (slot, kernel id, algorithmic FLOPs) of every instrumented GEMM / conv launch."""

    def __init__(self):
        from sbl_for_multilingual_lip_reading_amd import _lib, ops
        self.ops, self.lib, self.active, self.launches = ops, _lib.load(), False, []
        inner = ops.call

        def call(name, *a):
            if not self.active:
                return inner(name, *a)
            s0 = self.lib.sbl_profile_used()       # process-wide counter: backward runs on the autograd thread
            inner(name, *a)
            desc = name
            if name == "sbl_gemm_f32":
                fl = 2.0 * a[2] * a[3] * a[4]
                by = 4.0 * (a[2] * a[4] + a[3] * a[4] + a[2] * a[3])
                desc = "gemm ta%d tb%d M%d N%d K%d" % (a[0], a[1], a[2], a[3], a[4])
            elif name == "sbl_gemm2_f32":          # both decoder directions in one launch
                fl = 4.0 * a[0] * a[1] * a[2]
                by = 8.0 * (a[0] * a[2] + a[1] * a[2] + a[0] * a[1])
                desc = "gemm2 M%d N%d K%d" % (a[0], a[1], a[2])
            elif name == "sbl_wgrad_seg_f32":      # merged decoder weight gradient: contracts over all stages' rows
                fl = 2.0 * a[6] * a[7] * sum(a[5][i] for i in range(a[0]))
                by = 4.0 * ((a[6] + a[7]) * sum(a[5][i] for i in range(a[0])) + 2 * a[6] * a[7])
            elif name == "sbl_wgrad_group_f32":    # every deferred weight gradient in one launch
                rows = sum(a[2][i] for i in range(a[1]))
                fl = sum(2.0 * a[7][p] * a[8][p] * rows for p in range(a[0]))
                by = sum(4.0 * ((a[7][p] + a[8][p]) * rows + 2 * a[7][p] * a[8][p]) for p in range(a[0]))
            elif name == "sbl_conv1x1s2_dgrad_compact":      # the downsample branch's input gradient on its even/even support
                nimg, h, w, cin, cout = a[3:8]
                fl = 2.0 * nimg * ((h + 1) // 2) * ((w + 1) // 2) * cout * cin
                by = 4.0 * (nimg * ((h + 1) // 2) * ((w + 1) // 2) * (cin + cout) + cout * cin)
                desc = "conv2d_dgrad(1x1/s2 compact) n%d %dx%d c%d->%d" % (nimg, h, w, cin, cout)
            elif name in ("sbl_conv2d_fwd", "sbl_conv2d_dgrad", "sbl_conv2d_dgrad_bnstats", "sbl_conv2d_dgrad_fused", "sbl_conv2d_wgrad"):
                off = 2 if name == "sbl_conv2d_fwd" else 0
                nimg, h, w, cin, cout, kh, kw, stride, pad = a[3 + off:12 + off]
                ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
                fl = 2.0 * nimg * ho * wo * cout * kh * kw * cin
                by = 4.0 * (nimg * h * w * cin + nimg * ho * wo * cout + cout * kh * kw * cin)
                desc = "%s n%d %dx%d c%d->%d k%d s%d" % (name[4:].replace("_bnstats", "").replace("_fused", ""), nimg, h, w, cin, cout, kh, stride)
            elif name == "sbl_stem_conv_fwd" or name == "sbl_stem_wgrad":
                n, t, h, w = a[4:8] if name == "sbl_stem_conv_fwd" else a[12:16]
                pix = n * t * (h // 2) * (w // 2)
                fl = 2.0 * pix * 64 * 245
                by = 4.0 * (n * t * h * w + pix * 64) + (0 if name == "sbl_stem_conv_fwd" else 4.0 * (pix // 4) * 64 + (pix // 4) * 64)
                desc = "%s n%d t%d %dx%d" % (name[4:], n, t, h, w)
            elif name == "sbl_stem_bn_relu_pool_fwd" or name == "sbl_stem_bwd_reduce":
                nt, ho, wo = a[7:10] if name == "sbl_stem_bn_relu_pool_fwd" else a[8:11]
                fl = 0.0
                by = 4.0 * nt * ho * wo * 64 + 5.0 * nt * (ho // 2) * (wo // 2) * 64
                desc = "%s nt%d %dx%d" % (name[4:], nt, ho, wo)
            elif name in ("sbl_attention_seg_fwd", "sbl_attention_seg_bwd"):
                o = 11 if name.endswith("fwd") else 15
                b_, h_, segs, nseg, lk = a[o], a[o + 1], a[o + 2], a[o + 3], a[o + 4]
                pairs = sum(segs[i] * (lk if lk > 0 else segs[i]) for i in range(nseg))
                fl = (4.0 if name.endswith("fwd") else 10.0) * b_ * h_ * pairs * 64      # QK^T + PV; five products in backward
                by = 4.0 * b_ * h_ * (pairs + 64 * sum(segs[i] for i in range(nseg)) * 4)
                desc = "%s B%d H%d L%s Lk%d" % (name[4:], b_, h_, [segs[i] for i in range(nseg)], lk)
            else:
                return
            s1 = self.lib.sbl_profile_used()
            if s1 > s0:      # a call may make several launches (stride-2 input gradients: one per parity class)
                self.launches.append((tuple(range(s0, s1)), self.lib.sbl_profile_last_kernel(), fl, desc, by))
        ops.call = call


def cpu_baseline(batch, T, hw):
    """CPU oracle fwd + loss + bwd on `batch` clips of the same synthetic workload and mode as the GPU step (train-mode
    BatchNorm, dropout ON: 0.1 in the transformer and the always-on 0.5 mask on the frontend features), SURVEY 8d: one
    warm-up and two timed steps on every core of this process's affinity mask (capped at 16: os.cpu_count() reports the
    whole host and oversubscribed OpenMP crawls), then one timed step on 8 threads for comparison with the survey's
    8-vCPU probe of the reference itself (1.27 clips/s)."""
    from oracle import sbl_oracle as O
    from sbl_for_multilingual_lip_reading_amd import detfill
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    sd = O.make_state_dict(6, 6, requires_grad=True)
    x, l2r, r2l = detfill.synthetic_batch(batch, T, hw, hw, 7)
    random.seed(7)
    coins = O.draw_coins()
    xt, lt, rt = torch.from_numpy(x), torch.from_numpy(l2r), torch.from_numpy(r2l)
    g = torch.Generator().manual_seed(7)

    def step():
        mask = (torch.rand(batch * T, 512, generator=g) >= 0.5).float()
        t0 = time.time()
        out = O.transformer_forward(sd, xt, lt, rt, coins, drop=0.1, frontend_mask=mask)
        O.train_step_loss(out).backward()
        for v in sd.values():
            v.grad = None
        return time.time() - t0

    torch.set_num_threads(cores)
    step()                                        # warm-up (allocator, oneDNN primitive caches)
    dts = [step(), step()]
    dt = sum(dts) / len(dts)
    dt8 = None
    if cores != 8:
        torch.set_num_threads(min(8, cores))
        dt8 = step()
        torch.set_num_threads(cores)
    return {"value": round(batch / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "value_8_threads": None if dt8 is None else round(batch / dt8, 4),
            "sample": "oracle/sbl_oracle.py full SBL 6+6 fwd+loss+bwd, dropout on, BN train: 1 warm-up + 2 timed steps of %d clips %dx%dx%d "
                      "fp32 on %d threads (%.1f s per step)%s, %s"
                      % (batch, T, hw, hw, cores, dt, "" if dt8 is None else "; 1 step on 8 threads (%.1f s)" % dt8, _cpu_model())}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def main():
    args = parse()
    if args.hang_dump > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_dump, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # one rank per GPU; wraps only in single-GPU rehearsals (gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from sbl_for_multilingual_lip_reading_amd import _lib, detfill, dp, ops
    from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
    lib = _lib.load()          # fail loudly if the HIP library is missing
    ops.set_matmul_precision(args.precision)
    ops.PACK_CACHE = not args.no_pack_cache
    for kv in args.tuning:
        ops.call("sbl_set_tuning", *[int(v) for v in kv.split("=")])
    rec = LaunchRecorder()

    B = args.batch
    model = build_model(dev, not args.no_dropout)
    flat = dp.FlatModel(model)
    dp.broadcast_parameters(flat)
    exchange = dp.GradientExchange(flat, world, overlap=False)

    # rank r gets its own shard of the synthetic minibatch (weak scaling: 32 clips per GPU)
    x_np, l2r_np, r2l_np = detfill.synthetic_batch(B, args.T, args.HW, args.HW, 7 + rank)
    x = torch.from_numpy(x_np).to(dev)
    l2r, r2l = torch.from_numpy(l2r_np).to(dev), torch.from_numpy(r2l_np).to(dev)
    # teacher-forcing coins (decoder.py:176): the decoder batches the steps of each teacher-forced run, so the
    # launch sequence depends on the coin pattern.  A few patterns are drawn (same seed on every rank),
    # one hipGraph is captured per pattern and the timed loop cycles through them.
    patterns = draw_coin_patterns(args.coin_patterns)
    for kv in args.set:
        k_, v_ = kv.split("=", 1)
        tgt = ops if k_.startswith("ops.") else model.decoder
        setattr(tgt, k_.split(".")[-1], int(v_) if v_.lstrip("-").isdigit() else v_)
    model.decoder.two_streams = not args.single_stream
    model.decoder.batch_teacher_runs = not args.per_step_decoder
    drop = ops.dropout_state(dev)
    loss_out = torch.zeros((), device=dev)

    def fwd_bwd():
        drop.begin_step()
        flat.zero_grad()
        if args.workload == "frontend":      # BASELINE config 2
            feats = model.visual_frontend(x.unsqueeze(1))
            loss = feats.square().mean()
        else:
            pl, gl, pr, gr = model(x, l2r, r2l)
            loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
        loss.backward()
        ops.join_side_streams()          # the decoder's second stream rejoins before the step ends
        loss_out.copy_(loss.detach())

    # The same step in two parts (Transformer.forward, transformer.py:27-36, with the tape cut at the frontend features):
    # part A = forward + loss + backward of decoder and encoder, leaving d(loss)/d(features) in feats_d.grad;
    # part B = backward of the visual frontend.  Between them the decoder / encoder gradient segments are complete,
    # so their all-reduces run beside part B (overlap the exchange with the frontend backward).
    split = {}

    def part_a():
        drop.begin_step()
        flat.zero_grad()
        feats = model.visual_frontend(x.unsqueeze(4).permute(0, 4, 1, 2, 3))
        feats_d = feats.detach().requires_grad_(True)
        lengths = [feats_d.size(1)] * feats_d.size(0)
        enc, *_ = model.encoder(feats_d, lengths)
        pl, gl, pr, gr = model.decoder(l2r, r2l, enc, lengths)
        loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
        loss.backward()
        ops.join_side_streams()
        loss_out.copy_(loss.detach())
        split["feats"], split["dfeats"] = feats, feats_d.grad

    def part_b():
        split["feats"].backward(split["dfeats"])
        ops.join_side_streams()
        split.clear()

    def set_coins(i):
        model.decoder.coins_host = patterns[i % len(patterns)]

    # eager warm-up on the stream the graph will be captured on (its split-K workspaces get created here)
    cap_stream = torch.cuda.Stream()
    cap_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap_stream):
        for w in range(max(args.warmup, 1)):
            set_coins(w)
            fwd_bwd()
            exchange.finish()
    torch.cuda.synchronize()
    log(args, "eager warm-up done")

    use_split = args.workload == "full" and not args.no_graph and args.mode != "eager" and (args.split_graph == "on" or (args.split_graph == "auto" and world > 1))
    graph = None
    graphs = []
    # With a process group alive other threads (collective watchdog) may touch the runtime while this thread captures:
    # only this thread's calls are checked then.  Backward's launches come from the autograd thread either way.
    cap_mode = "thread_local" if world > 1 else "global"

    def capture(fn, pool=None):
        if os.environ.get("SBL_BENCH_FAIL_CAPTURE"):      # test knob for the eager fallback below
            raise RuntimeError("capture failure forced by SBL_BENCH_FAIL_CAPTURE")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, stream=cap_stream, capture_error_mode=cap_mode):
            fn()
        return g

    if not args.no_graph and args.mode != "eager":
        try:
            for i in range(len(patterns)):
                set_coins(i)
                if use_split:
                    try:
                        ga = capture(part_a)
                        gb = capture(part_b, pool=ga.pool())      # reads part A's saved activations
                        graphs.append((ga, gb))
                        continue
                    except Exception as e:      # keep the run alive: one graph per step, exchange after the replay
                        print("[bench] two-graph capture failed (%s: %s); falling back to one graph per step" % (type(e).__name__, e),
                              file=sys.stderr, flush=True)
                        split.clear()
                        use_split = False
                        graphs = []
                        torch.cuda.synchronize()
                        for j in range(i):      # re-capture the earlier patterns as single graphs
                            set_coins(j)
                            graphs.append(capture(fwd_bwd))
                        set_coins(i)
                graphs.append(capture(fwd_bwd))
            graph = graphs[0]
            log(args, "capture + instantiate done (%d coin patterns%s)" % (len(graphs), ", two graphs per step" if use_split else ""))
        except Exception as e:
            # last resort: the eager step (launch-bound by ~1 % only since the stage-batched decoder backward), with the
            # gradient exchange launched from tensor hooks inside backward
            print("[bench] graph capture failed (%s: %s); running the eager step" % (type(e).__name__, e), file=sys.stderr, flush=True)
            split.clear()
            graph, graphs, use_split = None, [], False
        torch.cuda.synchronize()
    step_no = [0]

    def step_graph():
        i = step_no[0]
        step_no[0] += 1
        if use_split:
            ga, gb = graphs[i % len(graphs)]
            ga.replay()
            exchange.launch("decoder.")      # side stream: runs beside the frontend backward below
            exchange.launch("encoder.")
            gb.replay()
            exchange.finish()                # frontend segment + join
        else:
            graphs[i % len(graphs)].replay()
            exchange.finish()

    eager_exchange = [None]

    def step_eager():
        i = step_no[0]
        step_no[0] += 1
        if eager_exchange[0] is None:        # decoder / encoder segments are all-reduced from hooks inside backward
            eager_exchange[0] = dp.GradientExchange(flat, world, overlap=True)
        set_coins(i)
        with torch.cuda.stream(cap_stream):
            fwd_bwd()
            eager_exchange[0].finish()

    host_issue_ms = {}

    def trial(fn, n):
        """ms per step over n steps, max over ranks (every rank must reach the same decision); also records how long the
        host took to ISSUE a step (max over ranks): an eager step is only GPU-bound while that stays well below the step time"""
        step_no[0] = 0
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        issue = (time.perf_counter() - t) / n * 1e3
        log(args, "  host side of %s: %.2f ms/step to issue" % (fn.__name__, issue))
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t) / n * 1e3, issue], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        host_issue_ms[fn.__name__.replace("step_", "")] = round(float(t[1].item()), 2)
        return float(t[0].item())

    mode = "eager" if (args.no_graph or graph is None) else args.mode
    trial_ms = {}
    if mode == "auto":
        n_trial = 2 * len(patterns)
        for name, fn in (("graph", step_graph), ("eager", step_eager)):
            try:
                trial_ms[name] = trial(fn, n_trial)
            except Exception as e:       # keep the run alive with the other way of issuing the step
                print("[bench] %s trial failed (%s: %s)" % (name, type(e).__name__, e), file=sys.stderr, flush=True)
                torch.cuda.synchronize()
        if not trial_ms:
            raise SystemExit("bench: neither graph replay nor the eager step runs")
        # Graph replay is the default.  Eager launches are taken when they are faster over the 16 trial steps (on hosts that
        # issue a step in ~60 % of its GPU time they are, by 1-1.5 %: the replayed multi-stream graph overlaps its branches a
        # little less than the streams do) AND the host has slack: an eager step whose issue time is close to the step time is
        # bound by the host's launch rate, which a trial flatters while the launch queue fills (39 ms in an 8-step trial, 55 ms
        # timed, measured on one box in round 2) and which N processes on one node share.
        mode = "graph" if "graph" in trial_ms else "eager"
        if "graph" in trial_ms and "eager" in trial_ms:
            if trial_ms["eager"] < 0.995 * trial_ms["graph"] and host_issue_ms.get("eager", 1e9) <= 0.8 * trial_ms["eager"]:
                mode = "eager"
        log(args, "trial: %s (host issue %s) -> %s" % (", ".join("%s %.2f ms" % kv for kv in trial_ms.items()), host_issue_ms, mode))
    if mode == "eager":
        graph, use_split = None, False
        if eager_exchange[0] is None:
            eager_exchange[0] = dp.GradientExchange(flat, world, overlap=True)
        exchange = eager_exchange[0]
    elif eager_exchange[0] is not None:
        eager_exchange[0].close()            # no collectives from hooks while graphs replay
    step = step_graph if mode == "graph" else step_eager
    step_no[0] = 0

    step()
    torch.cuda.synchronize()
    log(args, "first timed-style step done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ar_steps = [args.steps]
    if world > 1:
        exchange.launches.clear()        # all-reduce launches of the timed steps only (reported in `config`)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    loss_val = float(loss_out.item())
    log(args, "timed region done: %.2f ms/step" % (dt / args.steps * 1e3))

    # ---- per-kernel GPU time inside replays of the same step (in-kernel stamps, see docstring)
    fam = {}
    if rank == 0 and not args.no_kernel_timing:
        CAP = 32768
        init = torch.zeros(CAP, 2, dtype=torch.int64, device=dev)
        init[:, 0] = -1                                   # as uint64: +inf for atomicMin
        stamps = init.clone()
        exchange.world = 1          # rank 0 only from here on: no collectives (the eager fallback launches them from hooks)
        lib.sbl_profile_begin(stamps.data_ptr(), CAP)
        rec.active = True
        # the pattern with the expected number of own-arg-max coins (8): family times of a representative step
        set_coins(min(range(len(patterns)), key=lambda i_: (abs(sum(patterns[i_]) - 8), i_)))
        if graph is not None:
            pgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(pgraph, stream=cap_stream):
                fwd_bwd()
            run = pgraph.replay
        else:
            def run():
                with torch.cuda.stream(cap_stream):
                    fwd_bwd()
            run()
        rec.active = False
        torch.cuda.synchronize()
        used = lib.sbl_profile_end()
        launches = [l for l in rec.launches if l[0][-1] < used]
        if graph is not None:
            reps = 3
            dur = np.zeros(CAP)
            for _ in range(reps):
                stamps.copy_(init)
                torch.cuda.synchronize()
                run()
                torch.cuda.synchronize()
                s = stamps.cpu().numpy()
                dur += (s[:, 1] - s[:, 0]) / 100.0        # 100 MHz ticks -> microseconds
            dur /= reps
        else:
            # eager: the stamps of the recorded pass itself (slot order is only defined within one pass)
            s = stamps.cpu().numpy()
            dur = (s[:, 1] - s[:, 0]) / 100.0
        for slots, kid, fl, _d, by in launches:
            f = fam.setdefault(kid, {"launches": 0, "us": 0.0, "flops": 0.0, "bytes": 0.0})
            f["bytes"] += by
            f["launches"] += len(slots)
            f["us"] += float(sum(dur[sl] for sl in slots))
            f["flops"] += fl
        if args.dump_launches:
            agg = {}
            for slots, kid, fl, d, _by in launches:
                e = agg.setdefault((kid, d), [0, 0.0, 0.0])
                e[0] += 1; e[1] += float(sum(dur[sl] for sl in slots)); e[2] += fl
            with open(args.dump_launches, "w") as f:
                for (kid, d), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write("%8.1f us  n=%3d  avg %7.1f us  %6.1f TF  kid %d  %s\n" % (us, n, us / n, fl / us / 1e6, kid, d))
        log(args, "kernel timing done (%d instrumented launches per step)" % len(launches))

    # ---- step time per coin pattern (outside the timed region): the step is linear in the number of own-arg-max coins (one
    # more sequential decoder stage each), so the fit also says what the round-2 pattern set (mean 7.25 coins) would read
    per_pattern = None
    if rank == 0 and world == 1 and graph is not None and not use_split and len(graphs) > 1:
        per_pattern_ms = []
        for g_ in graphs:
            g_.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            g_.replay()
            g_.replay()
            torch.cuda.synchronize()
            per_pattern_ms.append((time.perf_counter() - t1) / 2 * 1e3)
        cn = np.array([sum(p_) for p_ in patterns], dtype=np.float64)
        slope, icpt = np.polyfit(cn, np.array(per_pattern_ms), 1) if len(set(cn.tolist())) > 1 else (0.0, per_pattern_ms[0])
        per_pattern = {"own_argmax_coins": [int(c) for c in cn], "ms": [round(v, 3) for v in per_pattern_ms],
                       "ms_per_own_argmax_coin": round(float(slope), 3),
                       "ms_at_7.25_coins": round(float(icpt + 7.25 * slope), 3),      # the mean of round 2's four raw draws
                       "ms_at_8_coins": round(float(icpt + 8.0 * slope), 3)}
        log(args, "per-pattern timing done")

    # ---- the like-for-like exact-fp32 number, timed in this same run: the matmul precision is baked
    # into a hipGraph at capture, so the step is re-captured under "f32" for every coin pattern and replayed once per
    # pattern after a warm-up replay (rank 0 / N = 1 only; the headline `value` above is untouched by it)
    f32_exact = None
    if rank == 0 and world == 1 and graph is not None and args.precision == "bf16x6" and args.workload == "full" and not args.no_f32_exact:
        try:
            ops.set_matmul_precision("f32")
            fgraphs = []
            for i in range(len(patterns)):
                set_coins(i)
                fgraphs.append(capture(fwd_bwd))
            fgraphs[0].replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for g_ in fgraphs:
                g_.replay()
            torch.cuda.synchronize()
            fdt = (time.perf_counter() - t1) / len(fgraphs)
            f32_exact = {"ms_per_step": round(fdt * 1e3, 3), "value": round(B / fdt, 3), "unit": "clips/s", "steps": len(fgraphs),
                         "matmul_precision": "f32", "note": "same step, same coin patterns, v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain)"}
            del fgraphs
        finally:
            ops.set_matmul_precision(args.precision)
        log(args, "f32_exact done")

    if rank == 0:
        clips = B * world * args.steps
        timed_coins = [sum(patterns[i % len(patterns)]) for i in range(1, args.steps + 1)] if mode in ("graph", "eager") else []
        n_ar = len(exchange.launches) if world > 1 else 0
        out = {
            "metric": "lip-clips/sec fwd+bwd (%dx%dx%d)" % (args.T, args.HW, args.HW), "value": round(clips / dt, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": MODE_DTYPE[args.precision], "data": "synthetic",
            "config": {"workload": {"full": "full SBL 6+6 (Conv3d stem + ResNet-18 + encoder + SBL decoder) fwd+loss+bwd (BASELINE config 3)",
                                    "frontend": "visual frontend only (Conv3d stem + ResNet-18) fwd+bwd (BASELINE config 2)",
                                    "config5": "full SBL 6+6 on long clips, mixed bf16 (BASELINE config 5)"}[args.workload],
                       "matmul_precision": args.precision,
                       "per_gpu_batch": B, "global_batch": B * world, "clip": "%dx%dx%d" % (args.T, args.HW, args.HW), "parallelism": "dp%d" % world,
                       "dropout": not args.no_dropout, "bn": "train", "issue": mode, "trial_ms": trial_ms or None, "trial_host_issue_ms": host_issue_ms or None, "hipgraph": graph is not None, "graphs_per_step": 2 if (graph is not None and use_split) else 1,
                       "decoder_streams": 1 if args.single_stream else 2, "conv_weight_pack": "per step" if args.no_pack_cache else "cached (re-packed when weights change)", "ab_overrides": (args.set + ["tuning " + t for t in args.tuning]) or None,
                       "decoder_schedule": "per-step" if args.per_step_decoder else "teacher-forced runs batched",
                       "coin_patterns": len(patterns), "own_argmax_coins": [sum(p_) for p_ in patterns],
                       "mean_own_argmax_coins": round(sum(timed_coins) / max(len(timed_coins), 1), 3),      # over the timed steps; expectation 8
                       "backend": ("%s (%s)" % (args.backend, "RCCL over xGMI" if args.backend == "nccl" else "rehearsal")) if world > 1 else None,
                       "allreduce_launches_per_step": round(n_ar / max(ar_steps[0], 1), 2) if world > 1 else 0,
                       "allreduce_bytes_per_step": int(4 * sum(n for _, n in exchange.launches) / max(ar_steps[0], 1)) if world > 1 else 0,
                       "loss": round(loss_val, 5)},
        }
        if per_pattern is not None:
            out["config"]["per_pattern"] = per_pattern
        if f32_exact is not None:
            out["f32_exact"] = f32_exact
        if fam:
            peak = MODE_PEAK_TFLOPS[args.precision]
            fams = []
            for kid, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
                tf = f["flops"] / (f["us"] * 1e-6) / 1e12
                e = {"kid": kid, "bytes": f["bytes"], "kernel": KERNEL_NAMES.get(kid, str(kid)), "launches_per_step": f["launches"],
                     "ms_per_step": round(f["us"] / 1e3, 3), "avg_launch_us": round(f["us"] / f["launches"], 2),
                     "achieved_TFLOPs": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}
                if kid in (8, 9):      # the stem is also judged against HBM; attention is fp32 MFMA in every mode
                    e["frac"] = e["frac_of_fp32_mfma_peak"] if kid == 9 else round(tf / peak, 4)
                    e["achieved_TBs"] = round(f["bytes"] / (f["us"] * 1e-6) / 1e12, 3)
                    e["frac_of_hbm_peak"] = round(e["achieved_TBs"] / HBM_PEAK_TBS, 4)
                else:
                    e["frac"] = round(tf / peak, 4)
                fams.append(e)
            mfma = [f for f in fams if f["kid"] not in (8, 9)]
            top = mfma[0]
            out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved_TFLOPs"],
                               "peak": round(peak, 1), "unit": "TFLOP/s", "frac": top["frac"],
                               "peak_note": "fp32-equivalent TFLOP/s; peak = %s" % (
                                   "157.3 (v_mfma_f32_32x32x2_f32)" if args.precision == "f32" else
                                   "2500 (dense bf16 MFMA) / %d bf16 products per fp32 product" % {"bf16x6": 6, "bf16x3": 3, "bf16": 1}[args.precision]),
                               "frac_of_fp32_mfma_peak": top["frac_of_fp32_mfma_peak"],
                               "traffic": pmc_traffic(top["kid"]), "traffic_unit": "bytes/launch",
                               "traffic_source": "profiles/" + os.path.basename(PMC_SUMMARY) + " (this round's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_passes.sh)",
                               "algorithmic_bytes_per_launch": round(top["bytes"] / top["launches_per_step"]),
                               "avg_launch_us": top["avg_launch_us"],
                               "launches_per_step": top["launches_per_step"], "ms_per_step": top["ms_per_step"],
                               "families": [{k: v for k, v in f.items() if k not in ("kid", "bytes")} for f in fams if f is not top]}
        else:
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.T, args.HW)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()               # rank 0's instrumented replays above: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
