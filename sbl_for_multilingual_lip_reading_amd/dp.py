"""Data-parallel plumbing for the SBL hot path: one process per GPU, persistent replicas, one gradient
exchange per step.

The reference wraps the model in single-process nn.DataParallel (SBL/train.py:115): per step it re-broadcasts
324 MB of parameters from GPU0 and reduce-adds 324 MB of gradients back to GPU0 (hub and spoke).  Here every rank
keeps its replica; parameters and gradients live in two flat fp32 buffers, so the exchange is a handful of large
RCCL all-reduces over xGMI (no per-tensor launches, no parameter broadcast), issued on a side stream in ~25 MB
buckets in reverse-autograd order (decoder -> encoder -> ResNet stages 4..2 -> stage 1 + stem) so that all but the
last 0.6 MB travels beside the remaining backward.  BatchNorm statistics stay per replica, like the reference (no SyncBN).
"""
import torch
import torch.distributed as dist


def _ordered_params(model):
    """Parameters in registration order, except that
      * each MultiHeadAttention's (w_qs, w_ks, w_vs) weights and biases are emitted as adjacent triples - the fused-QKV
        GEMM needs them to be rows of one matrix;
      * the cross-attention K/V projections of ALL decoder layers (l2r layers 0.., then r2l layers 0..; decoder.py:41-53)
        are emitted as ONE block [w_ks_0; w_vs_0; w_ks_1; ...] (and one block of their biases): the K/V of the encoder
        output for every layer and direction are then one (N*T, 12*1024) GEMM in forward, and their input gradient one
        K = 12*1024 GEMM in backward (decoder_stages.py)."""
    from .transformer.attention import MultiHeadAttention
    seen, order = set(), []
    fused = []
    cross = []
    dec = getattr(model, "decoder", None)
    if dec is not None and hasattr(dec, "_layers"):
        cross = [lay.enc_attn for d in (0, 1) for lay in dec._layers(d)]
    cross_ids = {id(m) for m in cross}
    if cross:
        fused.append([w for m in cross for w in (m.w_ks.weight, m.w_vs.weight)])
        fused.append([b for m in cross for b in (m.w_ks.bias, m.w_vs.bias)])
    for mod in model.modules():
        if isinstance(mod, MultiHeadAttention) and id(mod) not in cross_ids:
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

    SEGMENTS: name prefixes in the order their gradients complete during backward (reverse autograd order: decoder,
    encoder, then the ResNet stages from the last to the first, the Conv3d stem at the very end); each becomes one
    contiguous range of the flat buffers, all-reduced in buckets of <= GradientExchange.bucket_bytes as soon as backward
    has passed it.  Only layer1 + stem (0.6 MB) are left to exchange after the step."""

    SEGMENTS = ("decoder.", "encoder.", "visual_frontend.resnet18.layer4.", "visual_frontend.resnet18.layer3.",
                "visual_frontend.resnet18.layer2.", "visual_frontend.")

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
                    p._sbl_flat = self
                    off += pad(n)
                self.ranges[seg] = (start, off)
        self.numel = total
        self.device_index = dev.index if dev.type == "cuda" else None
        self._in_backward = False
        _ops().register_flat_model(self)

    def zero_grad(self):
        self.flat_grad.zero_()

    def begin_backward(self):
        """Called by the FIRST tape node of every backward (ops._backward_enter; every sbl autograd Function's backward
        goes through it), i.e. before any kernel of that backward has accumulated into the flat gradient: this is the only
        point where a dropped `.grad` may be turned into a zeroed slice.  The reference loop zeroes between forward and
        backward (SBL/train.py:195-196: forward, optimizer.zero_grad(), loss.backward()), so the tape nodes' forward-time
        view of the buffers must stay valid and nothing may be zeroed once accumulation is under way."""
        self._in_backward = True
        torch.autograd.Variable._execution_engine.queue_callback(self._end_backward)
        if any(p.grad is not p._sbl_grad for p, _, _ in self.slots):
            self.reattach()

    def _end_backward(self):
        self._in_backward = False

    def reattach(self):
        """Repair after a foreign `zero_grad(set_to_none=True)` (torch.optim's default) or an assignment to `p.grad`: the
        kernels accumulate into the flat gradient whatever `p.grad` says, so a parameter whose `.grad` was dropped gets
        its slice zeroed (= the fresh gradient the caller asked for) and `.grad` pointed back at it; an assigned foreign
        gradient is copied into the slice first.  Runs at the root of a backward (begin_backward), never after a kernel of
        that backward has accumulated."""
        dropped = [(p, off) for p, off, _ in self.slots if p.grad is None]
        with torch.no_grad():
            if len(dropped) == len(self.slots):
                self.flat_grad.zero_()                       # the common case: one fill
            else:
                for p, off in dropped:
                    self.flat_grad[off:off + p.numel()].zero_()
            for p, off, _ in self.slots:
                view = self.flat_grad[off:off + p.numel()].view(p.shape)
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
                p._sbl_grad = view

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

    def span(self, prefix):
        """[lo, hi) of the flat buffers covering every segment whose name starts with `prefix` (segments are laid out in
        SEGMENTS order, so e.g. "visual_frontend." spans the four frontend segments)."""
        r = [self.ranges[s] for s in self.SEGMENTS if s.startswith(prefix)]
        return min(a for a, _ in r), max(b for _, b in r)


def _ops():
    from . import ops
    return ops


class GradientExchange:
    """Averages (average=True) or sums the flat gradient over ranks with RCCL all-reduces on a side stream, one bucket
    of <= bucket_bytes at a time in reverse-autograd order (~13 x 25 MB), each segment launched from a tensor
    hook that fires when backward has passed it, so the exchange runs beside the remaining backward.  average=False pairs
    with FusedAdam(grad_scale=1/world): the 1/R then costs nothing (folded into the update)."""

    def __init__(self, flat: FlatModel, world_size: int, overlap: bool = True, average: bool = True,
                 bucket_bytes: int = 25 << 20):
        self.flat = flat
        self.world = world_size
        self.average = average
        self.bucket = max(4, bucket_bytes // 4 // 4 * 4)          # elements per bucket, 16-byte aligned
        self.cuda = flat.flat_grad.is_cuda          # CPU tensors + gloo are used by the world_size-2 unit tests
        self.stream = torch.cuda.Stream() if (self.cuda and world_size > 1) else None
        self._hooks = []
        self._pending = []
        self.launches = []          # (segment, elements) per all-reduce issued, for tests / the bench line
        if world_size > 1 and overlap:
            self._install()

    def _install(self):
        """A segment's gradients are complete when backward reaches the gradient of the activation that FEEDS it: the
        encoder output for the decoder (all 16 steps and the hoisted K/V projections are behind it), the frontend
        features for the encoder, and the input of ResNet stage k for stage k.  Tensor hooks there launch the finished
        segment while the rest of backward runs.  (Parameter hooks cannot be used: the kernels accumulate into the flat
        gradient buffer themselves and autograd never sees those gradients.)"""
        model = self.flat.model

        def on_encoder_out(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if t.requires_grad:
                t.register_hook(lambda g: self.launch("decoder."))

        def on_frontend_out(mod, inp, out):
            if out.requires_grad:
                out.register_hook(lambda g: self.launch("encoder."))

        def stage_pre_hook(seg):
            def pre(mod, inp):
                x = inp[0]
                if torch.is_tensor(x) and x.requires_grad:
                    x.register_hook(lambda g: self.launch(seg))
            return pre

        self._hooks.append(model.encoder.register_forward_hook(on_encoder_out))
        self._hooks.append(model.visual_frontend.register_forward_hook(on_frontend_out))
        res = getattr(model.visual_frontend, "resnet18", None)
        if res is not None:
            for k in (4, 3, 2):
                self._hooks.append(getattr(res, "layer%d" % k).register_forward_pre_hook(
                    stage_pre_hook("visual_frontend.resnet18.layer%d." % k)))

    def close(self):
        """Remove the forward hooks (the exchange then only runs when launch() / finish() are called explicitly)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _all_reduce(self, seg, g):
        for a in range(0, g.numel(), self.bucket):
            b = g[a:a + self.bucket]
            if self.average:
                b.mul_(1.0 / self.world)
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
            self.launches.append((seg, b.numel()))

    def launch(self, seg, _from_finish=False):
        if self.world <= 1:
            return
        if seg in self._pending:
            if not _from_finish:
                # a second backward reached this segment before finish(): its first micro-batch was already averaged in
                # place, so accumulating on top of it would silently diverge the ranks
                raise RuntimeError("GradientExchange: segment %r was already exchanged in this step; call finish() after "
                                   "every backward (for gradient accumulation build the exchange with overlap=False and "
                                   "call finish() once after the last micro-batch)" % seg)
            return
        g = self.flat.segment_grad(seg)
        self._pending.append(seg)
        if g.numel() == 0:
            return
        if self.cuda:
            cur = torch.cuda.current_stream()
            # the finished segment's merged weight-gradient GEMMs (decoder and encoder layers defer theirs) must be
            # issued before its all-reduce; idempotent
            _ops().flush_deferred()
            side = _ops()._side_streams.get(cur.device_index)
            if side is not None:
                self.stream.wait_stream(side)       # trunk / deferred weight gradients are accumulated there
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self._all_reduce(seg, g)
        else:
            self._all_reduce(seg, g)

    def finish(self):
        """Call after backward: exchanges whatever has not gone out yet (layer1 + stem) and joins the side stream."""
        if self.world <= 1:
            return
        for seg in FlatModel.SEGMENTS:
            self.launch(seg, _from_finish=True)
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
