"""Data-parallel plumbing for the SBL hot path: one process per GPU, persistent replicas, one gradient
exchange per step.

The reference wraps the model in single-process nn.DataParallel (SBL/train.py:115): per step it re-broadcasts
324 MB of parameters from GPU0 and reduce-adds 324 MB of gradients back to GPU0 (hub and spoke).  Here every rank
keeps its replica; parameters and gradients live in two flat fp32 buffers, so the exchange is a handful of large
RCCL all-reduces over xGMI (no per-tensor launches, no parameter broadcast), issued on a side stream per segment
in reverse-autograd order (decoder -> encoder -> visual frontend) so the frontend's backward overlaps the
decoder's all-reduce.  BatchNorm statistics stay per replica, like the reference (no SyncBN).
"""
import torch
import torch.distributed as dist


def _ordered_params(model):
    """Parameters in registration order, except that each MultiHeadAttention's (w_qs, w_ks, w_vs) weights and
    biases are emitted as adjacent triples — the fused-QKV GEMM needs them to be rows of one matrix."""
    from .transformer.attention import MultiHeadAttention
    seen, order = set(), []
    fused = []
    for mod in model.modules():
        if isinstance(mod, MultiHeadAttention):
            fused.append([mod.w_qs.weight, mod.w_ks.weight, mod.w_vs.weight])
            fused.append([mod.w_qs.bias, mod.w_ks.bias, mod.w_vs.bias])
    lead = {id(g[0]): g for g in fused}
    member = {id(p) for g in fused for p in g}
    for p in model.parameters():
        if id(p) in seen:
            continue
        if id(p) in lead:
            for q in lead[id(p)]:
                seen.add(id(q))
                order.append(q)
        elif id(p) in member:
            continue          # emitted with its group leader
        else:
            seen.add(id(p))
            order.append(p)
    return order


class FlatModel:
    """Re-points every parameter (and its .grad) of `model` at a slice of one flat fp32 buffer.

    segments: name prefixes in the order their gradients complete during backward; each becomes one
    contiguous range of the flat buffers = one all-reduce."""

    SEGMENTS = ("decoder.", "encoder.", "visual_frontend.")

    def __init__(self, model):
        self.model = model
        names = {id(p): n for n, p in model.named_parameters()}
        params = _ordered_params(model)
        by_seg = {s: [] for s in self.SEGMENTS}
        for p in params:
            n = names[id(p)]
            seg = next((s for s in self.SEGMENTS if n.startswith(s)), self.SEGMENTS[-1])
            by_seg[seg].append(p)
        dev = params[0].device
        total = sum(p.numel() for p in params)
        # 16-byte alignment of every tensor start (float4 loads in the kernels): pad numel to a multiple of 4
        pad = lambda n: (n + 3) // 4 * 4
        total = sum(pad(p.numel()) for p in params)
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.ranges = {}
        self.slots = []          # (parameter, offset, padded numel) in buffer order
        off = 0
        with torch.no_grad():
            for seg in self.SEGMENTS:
                start = off
                for p in by_seg[seg]:
                    n = p.numel()
                    self.slots.append((p, off, pad(n)))
                    self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                    p.data = self.flat_param[off:off + n].view(p.shape)
                    p.grad = self.flat_grad[off:off + n].view(p.shape)
                    # ops.* backward kernels accumulate straight into this buffer (GEMM '+=' epilogues, atomics)
                    # and hand autograd None, so no AccumulateGrad add kernels run for these parameters
                    p._sbl_grad = p.grad
                    off += pad(n)
                self.ranges[seg] = (start, off)
        self.numel = total

    def zero_grad(self):
        self.flat_grad.zero_()

    def trainable_ranges(self):
        """Maximal contiguous [a, b) ranges of the flat buffers whose parameters have requires_grad=True (the
        reference's README stage 2 freezes the encoder and builds Adam over filter(requires_grad), SBL/train.py:75)."""
        out = []
        for p, off, n in self.slots:
            if not p.requires_grad:
                continue
            if out and out[-1][1] == off:
                out[-1][1] = off + n
            else:
                out.append([off, off + n])
        return [(a, b) for a, b in out]

    def segment_grad(self, seg):
        a, b = self.ranges[seg]
        return self.flat_grad[a:b]


def _ops():
    from . import ops
    return ops


class GradientExchange:
    """Averages the flat gradient over ranks: one RCCL all-reduce per segment on a side stream, launched from
    a hook that fires when the first parameter of the NEXT segment receives its gradient (i.e. the previous
    segment's backward is complete), overlapped with the remaining backward."""

    def __init__(self, flat: FlatModel, world_size: int, overlap: bool = True):
        self.flat = flat
        self.world = world_size
        self.cuda = flat.flat_grad.is_cuda          # CPU tensors + gloo are used by the world_size-2 unit tests
        self.stream = torch.cuda.Stream() if (self.cuda and world_size > 1) else None
        self._done = []
        self._hooks = []
        self._pending = []
        if world_size > 1 and overlap:
            self._install()

    def _install(self):
        """Backward reaches the gradient of the encoder output only after the whole decoder (all 16 steps and
        the hoisted K/V projections) has been differentiated, and the gradient of the frontend features only
        after the encoder: tensor hooks on those two activations launch the finished segment's all-reduce while
        the rest of backward is still running.  (Parameter hooks cannot be used: the kernels accumulate into the
        flat gradient buffer themselves and autograd never sees those gradients.)"""
        model = self.flat.model

        def on_encoder_out(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if t.requires_grad:
                t.register_hook(lambda g: self.launch("decoder."))

        def on_frontend_out(mod, inp, out):
            if out.requires_grad:
                out.register_hook(lambda g: self.launch("encoder."))

        self._hooks.append(model.encoder.register_forward_hook(on_encoder_out))
        self._hooks.append(model.visual_frontend.register_forward_hook(on_frontend_out))

    def close(self):
        """Remove the forward hooks (the exchange then only runs when launch() / finish() are called explicitly)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def launch(self, seg):
        if self.world <= 1:
            return
        g = self.flat.segment_grad(seg)
        if self.cuda:
            cur = torch.cuda.current_stream()
            # the finished segment's merged weight-gradient GEMMs (decoder and encoder layers defer theirs) must be
            # issued before its all-reduce; idempotent
            _ops().flush_deferred()
            side = _ops()._side_streams.get(cur.device_index)
            if side is not None:
                self.stream.wait_stream(side)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                g.mul_(1.0 / self.world)
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
        else:
            g.mul_(1.0 / self.world)
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        self._pending.append(seg)

    def finish(self):
        """Call after backward: exchanges the last segment and joins the side stream."""
        if self.world <= 1:
            return
        for seg in FlatModel.SEGMENTS:
            if seg not in self._pending:
                self.launch(seg)
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._pending = []


def broadcast_parameters(flat: FlatModel, src=0):
    """One broadcast at start-up replaces nn.DataParallel's per-step replicate()."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat.flat_param, src)
        for b in flat.model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, src)
