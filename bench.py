#!/usr/bin/env python
"""bench.py — lip-clips/s, forward+backward, of the SBL hot path on synthetic 29x88x88 clips (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                          # one rank per GPU, RCCL over xGMI

A "step" = one optimizer-free training step of the reference (SBL/train.py:188-196): Transformer.forward on a
batch of 32 clips per GPU (Conv3d stem -> ResNet-18 -> 6-layer encoder -> 16-step SBL decoder), the two
label-smoothed losses, loss.backward(), and for N > 1 the gradient average over ranks.  Dropout is ON (it is part
of the reference step), BatchNorm in training mode, fp32 throughout.  Inputs are synthetic and already resident in
HBM.  The step is captured once into a hipGraph (after the warm-up) and replayed; masks and coins still change per
replay because their seed / flags live in device memory.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — the dominant kernel of the step (by rocprofv3 time share, profiles/): HIP-event-timed live,
                 on the launch stream, over the timed region's own launches
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
# algorithmic work of the trunk's layer1 3x3 convolution (the single largest kernel):
# M = B*T*22*22 output pixels, N = 64, K = 576
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="clips per GPU")
    ap.add_argument("--workload", default="full", choices=["full", "frontend"])
    ap.add_argument("--no-graph", action="store_true", help="run eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--single-stream", action="store_true", help="run the two decoder directions on one stream")
    ap.add_argument("--hang-dump", type=int, default=0, help="debug: dump all Python stacks after this many seconds")
    ap.add_argument("--verbose", action="store_true", help="progress lines on stderr")
    return ap.parse_args()


def log(args, msg):
    if args.verbose:
        print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


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


class KernelTimer:
    """HIP events around one named C-ABI entry point, on the stream it is launched on (torch's current stream),
    collected inside the timed region."""

    def __init__(self, name, shape_filter=None):
        self.name, self.filter, self.events, self.enabled = name, shape_filter, [], False

    def install(self):
        from sbl_for_multilingual_lip_reading_amd import ops
        inner = ops.call

        def call(name, *args):
            if self.enabled and name == self.name and (self.filter is None or self.filter(args)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                inner(name, *args)
                b.record()
                self.events.append((a, b))
            else:
                inner(name, *args)
        ops.call = call

    def mean_ms(self):
        if not self.events:
            return None
        return float(np.mean([a.elapsed_time(b) for a, b in self.events]))


def cpu_baseline(batch):
    """CPU oracle fwd + loss + bwd on `batch` clips of the same synthetic workload (dropout neutralised: the
    oracle's dropout switch costs nothing relative to the convolutions)."""
    from oracle import sbl_oracle as O
    from sbl_for_multilingual_lip_reading_amd import detfill
    # the box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole host and
    # oversubscribing OpenMP by 10x makes the CPU path crawl, so use the affinity mask, capped at 16
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    sd = O.make_state_dict(6, 6, requires_grad=True)
    x, l2r, r2l = detfill.synthetic_batch(batch, T_FRAMES, HW, HW, 7)
    random.seed(7)
    coins = O.draw_coins()
    t0 = time.time()
    out = O.transformer_forward(sd, torch.from_numpy(x), torch.from_numpy(l2r), torch.from_numpy(r2l), coins)
    O.train_step_loss(out).backward()
    dt = time.time() - t0
    return {"value": round(batch / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/sbl_oracle.py full SBL 6+6 fwd+loss+bwd, one step of %d clips 29x88x88 fp32 (%.1f s), %s"
                      % (batch, dt, _cpu_model())}


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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from sbl_for_multilingual_lip_reading_amd import _lib, detfill, dp, ops
    from sbl_for_multilingual_lip_reading_amd.transformer.loss import cal_performance_device
    _lib.load()          # fail loudly if the HIP library is missing

    B = args.batch
    model = build_model(dev, not args.no_dropout)
    flat = dp.FlatModel(model)
    dp.broadcast_parameters(flat)
    exchange = dp.GradientExchange(flat, world, overlap=False)

    # rank r gets its own shard of the synthetic minibatch (weak scaling: 32 clips per GPU)
    x_np, l2r_np, r2l_np = detfill.synthetic_batch(B, T_FRAMES, HW, HW, 7 + rank)
    x = torch.from_numpy(x_np).to(dev)
    l2r, r2l = torch.from_numpy(l2r_np).to(dev), torch.from_numpy(r2l_np).to(dev)
    coins_dev = torch.zeros(16, dtype=torch.int32, device=dev)
    model.decoder.coins_dev = coins_dev
    model.decoder.two_streams = not args.single_stream
    rng = random.Random(7)            # same coin sequence on every rank
    drop = ops.dropout_state(dev)
    loss_out = torch.zeros((), device=dev)

    timer = KernelTimer("sbl_conv2d_fwd", lambda a: a[7] == 64 and a[8] == 64 and a[9] == 3)   # layer1 3x3 convs
    timer.install()

    def fwd_bwd():
        drop.begin_step()
        flat.zero_grad()
        if args.workload == "frontend":
            feats = model.visual_frontend(x.unsqueeze(1))
            loss = feats.square().mean()
        else:
            pl, gl, pr, gr = model(x, l2r, r2l)
            loss = 0.5 * (cal_performance_device(pl, gl, 0.1)[0] + cal_performance_device(pr, gr, 0.1)[0])
        loss.backward()
        ops.join_side_streams()          # the decoder's second stream rejoins before the step ends
        loss_out.copy_(loss.detach())

    def new_coins():
        coins_dev.copy_(torch.tensor([int(rng.random() > 0.5) for _ in range(16)], dtype=torch.int32), non_blocking=False)

    # eager warm-up (also first-touch of every kernel / attribute before capture)
    graph = None
    for w in range(max(args.warmup, 1)):
        new_coins()
        fwd_bwd()
        exchange.finish()
    torch.cuda.synchronize()
    log(args, "eager warm-up done")
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd_bwd()                    # one more eager run on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        log(args, "capture begin")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):     # same stream as the warm-up: its split-K workspace exists
            fwd_bwd()
        log(args, "capture + instantiate done")
        torch.cuda.synchronize()
        new_coins()
        graph.replay()
        exchange.finish()
        torch.cuda.synchronize()
        log(args, "first replay done")

    def step():
        new_coins()
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        exchange.finish()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = graph is None       # events cannot be recorded inside a replay; see below
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    loss_val = float(loss_out.item())

    # dominant kernel, timed live with HIP events on its launch stream.  When the step is replayed from a graph
    # the events cannot sit inside the replay, so the same kernel is re-launched on the same operands right after
    # the timed region (same process, same clocks), 20 launches, events around each.
    kernel_ms = timer.mean_ms()
    if kernel_ms is None and rank == 0:
        NT = B * T_FRAMES
        xa = torch.randn(NT, 22, 22, 64, device=dev)
        w = torch.randn(64, 3, 3, 64, device=dev)
        y = torch.empty(NT, 22, 22, 64, device=dev)
        st = torch.empty(128, device=dev, dtype=torch.float64)
        evs = []
        for i in range(25):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops._lib.call("sbl_conv2d_fwd", xa.data_ptr(), w.data_ptr(), y.data_ptr(), st.data_ptr(), NT, 22, 22, 64, 64, 3, 3, 1, 1,
                          torch.cuda.current_stream().cuda_stream)
            b.record()
            if i >= 5:
                evs.append((a, b))
        torch.cuda.synchronize()
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    if rank == 0:
        clips = B * world * args.steps
        flops = 2.0 * (B * T_FRAMES * 22 * 22) * 64 * 576          # algorithmic FLOPs of one layer1 conv launch
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None
        out = {
            "metric": "lip-clips/sec fwd+bwd (29x88x88)", "value": round(clips / dt, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("full SBL 6+6 (Conv3d stem + ResNet-18 + encoder + SBL decoder) fwd+loss+bwd"
                                    if args.workload == "full" else "visual frontend only (Conv3d stem + ResNet-18) fwd+bwd"),
                       "per_gpu_batch": B, "global_batch": B * world, "clip": "29x88x88", "parallelism": "dp%d" % world,
                       "dropout": not args.no_dropout, "bn": "train", "hipgraph": graph is not None, "loss": round(loss_val, 5)},
            "roofline": {"bound": "mfma", "kernel": "sbl_mfma_gemm_kernel<ConvGatherKC,DenseKC> (trunk layer1 conv3x3 fwd, "
                                                    "M=%d N=64 K=576)" % (B * T_FRAMES * 484),
                         "achieved": None if achieved is None else round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None, "kernel_ms": None if kernel_ms is None else round(kernel_ms, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
