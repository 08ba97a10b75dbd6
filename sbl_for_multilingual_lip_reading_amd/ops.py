"""torch.autograd Functions over the C ABI of libsbl_hip.so.

torch is plumbing only here: it owns device memory (caching allocator), the
current HIP stream and the autograd tape; every FLOP of forward and backward runs
in a hand-written HIP kernel reached through sbl_for_multilingual_lip_reading_amd._lib.call().
Nothing in this file falls back to torch math on failure: a missing library or a
non-zero status raises.

All Functions are hipGraph-capturable: no host sync, no host read of device data;
dropout seeds and teacher-forcing coins live in device memory.
"""
import torch

try:
    from . import _lib
except ImportError:      # drop-in mode: this directory itself is on sys.path
    import _lib

call = _lib.call

_PRECISIONS = {"f32": 0, "bf16x6": 6, "bf16x3": 3, "bf16": 1}


def set_matmul_precision(mode):
    """Arithmetic of every dense GEMM / trunk convolution (include/sbl_hip.h, sbl_set_matmul_precision), process-wide:
    "f32" exact fp32 MFMA; "bf16x6" exact three-way bf16 split, six bf16 MFMA products, fp32 accumulation (fp32-grade
    results, the default of bench.py); "bf16x3" two planes / three products; "bf16" plain bf16 inputs (BASELINE config 5)."""
    if mode not in _PRECISIONS:
        raise ValueError("matmul precision %r (one of %s)" % (mode, ", ".join(_PRECISIONS)))
    call("sbl_set_matmul_precision", _PRECISIONS[mode])


def get_matmul_precision():
    v = _lib.load().sbl_get_matmul_precision()
    return [k for k, t in _PRECISIONS.items() if t == v][0]


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


# One pooled memset per training step instead of one per convolution: the packed weight-gradient buffers (split-K float atomics
# start from zero) and the BatchNorm-backward sums that ride on input-gradient epilogues are carved from a per-device pool
# that StemFn.forward - the first node of a step - zeroes with a single launch.  A region is handed out at most once per
# reset, so it still holds zeros when its kernel runs; without a reset in this process, or when the pool is used up, the
# takers fall back to a buffer the C call zeroes itself.
ZERO_POOL_BYTES = 64 << 20        # ResNet-18 trunk weights: 44.7 MB of fp32, + 0.2 MB of fp64 sums
_ZERO_POOL = {}


def zero_pool_reset(dev):
    st = _ZERO_POOL.get(dev.index)
    if st is None:
        st = _ZERO_POOL[dev.index] = {"buf": torch.empty(ZERO_POOL_BYTES, dtype=torch.uint8, device=dev), "off": 0, "armed": False}
    st["buf"].zero_()
    st["off"], st["armed"] = 0, True


def zero_pool_take(dev, shape, dtype):
    """A zero-filled tensor from the pool, or None (the caller then allocates and lets the C call zero its buffer)."""
    st = _ZERO_POOL.get(dev.index)
    n = 1
    for d in shape:
        n *= d
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    if st is None or not st["armed"] or st["off"] + nbytes > ZERO_POOL_BYTES:
        return None
    off = st["off"]
    st["off"] = (off + nbytes + 255) & ~255
    return st["buf"][off:off + nbytes].view(dtype).view(*shape)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SblHipError("sbl ops need CUDA/HIP tensors (got a %s tensor); there is no CPU path" % t.device)


def _rows(t):
    """(M, ld) of a 2-D fp32 tensor whose last dim is dense."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32, (t.shape, t.stride(), t.dtype)
    return t.size(0), (t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1)))


# --------------------------------------------------------------------------- #
# dropout state: one device-resident seed shared by every mask of a step
# --------------------------------------------------------------------------- #
class DropoutState:
    """Device seed + per-call-site offsets.  `next_offset()` hands out a distinct
    stream offset per dropout site per forward; `bump()` advances the seed on the
    device (one tiny kernel), so a captured graph draws new masks on every replay."""

    def __init__(self, device, seed=None):
        if seed is None:
            # data-parallel replicas draw independent masks (nn.DataParallel's replicas each use their device's
            # generator, SBL/train.py:115): fold the rank into the default seed
            seed = 0x5B1C0FFEE
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                seed ^= (dist.get_rank() * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
        self.seed = torch.tensor([seed], dtype=torch.int64, device=device)
        self._offset = 0

    def next_offset(self):
        self._offset += 1
        return self._offset

    def begin_step(self):
        self._offset = 0
        call("sbl_seed_bump", _p(self.seed), _s())


_dropout_states = {}


def dropout_state(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _dropout_states.get(key)
    if st is None:
        st = _dropout_states[key] = DropoutState(device)
    return st


_side_streams = {}


def side_stream(device):
    """The per-device second HIP stream used to overlap the two decoder directions."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        # (stream priorities were measured and dropped: a low-priority side stream changes nothing, a high-priority
        # main stream captured into the hipGraph runs the step at 79 ms instead of 45)
        st = _side_streams[idx] = torch.cuda.Stream(device=idx)
    return st


_main_streams = {}


def set_main_stream(stream):
    """Remember which stream the side stream forks from / joins to (per device)."""
    _main_streams[stream.device_index] = stream


def _xs_in(*ts):
    """A tape node running on the side stream reads tensors that the main stream allocated (incoming grads):
    tell the caching allocator, or the block could be handed to a main-stream kernel while we still read it.
    Forward needs none of this (fork/join fences on both sides of every layer); backward's engine-inserted
    syncs are one-directional."""
    cur = torch.cuda.current_stream()
    if _side_streams.get(cur.device_index) is not None and cur == _side_streams[cur.device_index]:
        for t in ts:
            if t is not None:
                t.record_stream(cur)


def _xs_out(*ts):
    """...and what it hands back (allocated on the side stream) is consumed on the main stream."""
    cur = torch.cuda.current_stream()
    if _side_streams.get(cur.device_index) is not None and cur == _side_streams[cur.device_index]:
        main = _main_streams.get(cur.device_index)
        if main is not None:
            for t in ts:
                if t is not None:
                    t.record_stream(main)


_side_join_state = {}      # per device: nn.DataParallel drives one replica per device from its own thread


def _side_join_for_current_device():
    return _side_join_state.setdefault(torch.cuda.current_device(), {"armed": False})


def _arm_side_join():
    """Once per backward: make the stream that finishes backward wait for the side stream."""
    _side_join = _side_join_for_current_device()
    if not _side_join["armed"]:
        _side_join["armed"] = True

        def join():
            _side_join["armed"] = False
            cur = torch.cuda.current_stream()
            side = _side_streams.get(cur.device_index)
            if side is not None and side != cur:
                cur.wait_stream(side)
        torch.autograd.Variable._execution_engine.queue_callback(join)


def _side_to_main():
    """Explicit fence when a side-stream tape node hands its result to the main stream (belt and braces next to the
    autograd engine's own producer/consumer event)."""
    cur = torch.cuda.current_stream()
    if _side_streams.get(cur.device_index) is not None and cur == _side_streams[cur.device_index]:
        main = _main_streams.get(cur.device_index)
        if main is not None:
            main.wait_stream(cur)


class WgradCollector:
    """Deferred weight gradients of the decoder.  Every decoder weight is used once per stage (up to 16 times per
    step); instead of one skinny dW GEMM per use inside backward's dependency chain, the (dY, X) operand pairs are
    collected and each weight gets ONE GEMM that contracts over all stages' rows (K ~ 4352) when backward has
    finished (autograd engine callback).  Needs persistent gradient buffers (dp.FlatModel)."""

    def __init__(self):
        self.entries = {}
        self.armed = False
        self.flushes = 0

    def add(self, C, ldc, colsum, A, lda, B, ldb, rows, M, N):
        e = self.entries.get(C.data_ptr())
        if e is None:
            e = self.entries[C.data_ptr()] = {"C": C, "ldc": ldc, "colsum": colsum, "lda": lda, "ldb": ldb, "M": M,
                                              "N": N, "A": [], "B": [], "rows": []}
        e["A"].append(A)
        e["B"].append(B)
        e["rows"].append(rows)
        if not self.armed:
            self.armed = True
            self.device_index = torch.cuda.current_device()
            _armed.setdefault(self.device_index, []).append(self)
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)

    def flush(self):
        """Issue the collected GEMMs.  Called from a hook on the decoder's encoder_outputs gradient (every decoder
        tape node has run by then: they all outrank the hoisted K/V projections in the engine's ready queue), so
        the GEMMs go to the side stream and overlap the encoder / frontend backward on the main stream; called
        again from the engine's end-of-backward callback, which issues anything that arrived late and joins."""
        if not self.entries:
            return
        cur = torch.cuda.current_stream()
        side = _side_streams.get(cur.device_index)
        run = cur
        if side is not None and cur != side:
            cur.wait_stream(side)              # operands produced by the other direction's stream
            side.wait_stream(cur)
            run = side
        with torch.cuda.stream(run):
            # weights whose stages have the same row structure (all layer weights of both directions) form one group
            groups = {}
            for e in self.entries.values():
                groups.setdefault(tuple(e["rows"]), []).append(e)
            for rows0, ents in groups.items():
                grouped = (len(ents) > 1 and len(rows0) <= 16 and all(r % 16 == 0 for r in rows0)
                           and all(e["M"] % 4 == 0 and e["N"] % 4 == 0 for e in ents))
                if grouped:
                    # one launch for every weight: each 128x128 tile of each gradient is owned by one workgroup over
                    # the whole K = all stages' rows (no split-K, no atomics; see sbl_wgrad_group_f32)
                    n, k = len(ents), len(rows0)
                    need = _lib.load().sbl_wgrad_group_table_bytes(n)
                    key = (run.device_index, run.cuda_stream, len(groups) > 1 and rows0)
                    tab = _group_tables.get(key)
                    if tab is None or tab.numel() < need:
                        tab = _group_tables[key] = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=torch.device("cuda", run.device_index))
                    pa = (_ct.c_void_p * (n * k))(*[t.data_ptr() for e in ents for t in e["A"]])
                    pb = (_ct.c_void_p * (n * k))(*[t.data_ptr() for e in ents for t in e["B"]])
                    call("sbl_wgrad_group_f32", n, k, (_ct.c_int * k)(*rows0), pa, (_ct.c_long * n)(*[e["lda"] for e in ents]), pb,
                         (_ct.c_long * n)(*[e["ldb"] for e in ents]), (_ct.c_int * n)(*[e["M"] for e in ents]),
                         (_ct.c_int * n)(*[e["N"] for e in ents]), (_ct.c_void_p * n)(*[e["C"].data_ptr() for e in ents]),
                         (_ct.c_long * n)(*[e["ldc"] for e in ents]), (_ct.c_void_p * n)(*[_p(e["colsum"]) for e in ents]),
                         tab.data_ptr(), tab.numel(), _s())
                for e in ents:
                    if not grouped:
                        n = len(e["rows"])
                        for i in range(0, n, 16):
                            A, B, rows = e["A"][i:i + 16], e["B"][i:i + 16], e["rows"][i:i + 16]
                            k = len(rows)
                            pa = (_ct.c_void_p * k)(*[t.data_ptr() for t in A])
                            pb = (_ct.c_void_p * k)(*[t.data_ptr() for t in B])
                            pr = (_ct.c_int * k)(*rows)
                            call("sbl_wgrad_seg_f32", k, pa, e["lda"], pb, e["ldb"], pr, e["M"], e["N"], _p(e["C"]), e["ldc"],
                                 _p(e["colsum"]), _s())
                    for t in e["A"] + e["B"]:
                        t.record_stream(run)
        self.entries = {}
        self.flushes += 1

    def finish(self):
        """Engine end-of-backward callback: late entries, then the main stream waits for the side stream."""
        self.flush()
        cur = torch.cuda.current_stream()
        side = _side_streams.get(cur.device_index)
        if side is not None and cur != side:
            cur.wait_stream(side)
        self.armed = False
        lst = _armed.get(getattr(self, "device_index", -1), [])
        if self in lst:
            lst.remove(self)


_group_tables = {}
_armed = {}       # device index -> collectors with pending entries


def flush_deferred():
    """Issue every deferred weight-gradient GEMM collected so far (idempotent).  dp.GradientExchange calls this
    before it all-reduces the decoder segment; the decoder calls it from its encoder_outputs gradient hook."""
    for c in list(_armed.get(torch.cuda.current_device(), [])):
        c.flush()


import threading as _threading

_tls = _threading.local()      # the collector of the forward pass running on THIS thread (nn.DataParallel: one per replica)


def begin_defer():
    """Decoder forward: tape nodes created from here on defer their weight gradients (if enabled)."""
    _tls.collector = WgradCollector()


def end_defer():
    _tls.collector = None


def wgrad_gemm(M, N, K, A, lda, B, ldb, C, ldc, acc, colsum, defer=None):
    """dW (+)= A^T B with the bias gradient riding on it.  With a collector and a persistent gradient buffer (acc=1)
    the product is deferred to one all-stages GEMM per weight; else it is issued now."""
    if defer is not None and acc:
        defer.add(C, ldc, colsum, A, lda, B, ldb, K, M, N)
        return
    gemm(1, 0, M, N, K, A, lda, B, ldb, C, ldc, accumulate=acc, colsum=colsum)


def join_side_streams():
    """Make the current stream wait for everything enqueued on the side stream (end of a step)."""
    cur = torch.cuda.current_stream()
    side = _side_streams.get(cur.device_index)
    if side is not None:
        cur.wait_stream(side)


# split-K workspace: int[4096] tile counters (kept zero by the kernel) + fp32 partial slabs, one per stream so
# that GEMMs running concurrently on different streams never share slabs
WS_BYTES = 16 << 20
_workspaces = {}


def _workspace():
    st = torch.cuda.current_stream()
    key = (st.device_index, st.cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = torch.zeros(WS_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", st.device_index))
    return ws


def gemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, relu=0, mask=None, ldm=0, accumulate=0, colsum=None):
    ws = _workspace()
    call("sbl_gemm_f32", ta, tb, M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, _p(bias), relu, _p(mask), ldm,
         accumulate, _p(colsum), ws.data_ptr(), WS_BYTES, _s())


def gemm2(M, N, K, A0, A1, lda, B0, B1, ldb, C0, C1, ldc, bias0=None, bias1=None, relu=0):
    """C_d = A_d @ B_d^T (+ bias_d) (ReLU) for two same-shape problems in one launch (the two decoder directions)."""
    ws = _workspace()
    call("sbl_gemm2_f32", M, N, K, _p(A0), _p(A1), lda, _p(B0), _p(B1), ldb, _p(C0), _p(C1), ldc, _p(bias0), _p(bias1), relu,
         ws.data_ptr(), WS_BYTES, _s())


import ctypes as _ct

_seg_arrays = {}


def _segs(segL):
    """host int array of prefix lengths for the ragged ("segmented") kernels, cached per tuple"""
    segL = tuple(int(v) for v in segL)
    arr = _seg_arrays.get(segL)
    if arr is None:
        arr = _seg_arrays[segL] = (_ct.c_int * len(segL))(*segL)
    return arr, len(segL)


def _gbuf(p):
    """The persistent gradient buffer of a parameter, if the model was flattened (dp.FlatModel): backward then
    accumulates into it inside the kernels (GEMM epilogue '+=', atomics) and returns None to autograd, instead of
    allocating a gradient and having AccumulateGrad add it with a separate kernel (16x per decoder parameter).
    The buffer is the parameter's fixed slice of the flat gradient whatever `p.grad` currently says; a `.grad` the caller
    dropped or replaced (torch.optim's zero_grad(set_to_none=True) default, possibly BETWEEN forward and backward as in
    SBL/train.py:195-196) is repaired once, at the root of the next backward (_backward_enter)."""
    if p is None:
        return None
    return getattr(p, "_sbl_grad", None)


import functools as _functools
import weakref as _weakref

_flat_models = _weakref.WeakSet()


def register_flat_model(fm):
    _flat_models.add(fm)


def _backward_enter():
    """First thing every sbl tape node's backward does: the first node of a backward pass lets each flat model of this
    device repair / zero detached gradients BEFORE any kernel accumulates (dp.FlatModel.begin_backward)."""
    if not _flat_models:
        return
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    for fm in list(_flat_models):
        if not fm._in_backward and fm.device_index in (None, dev):
            fm.begin_backward()


def _bw(fn):
    @_functools.wraps(fn)
    def wrapped(ctx, *grads):
        _backward_enter()
        return fn(ctx, *grads)
    return wrapped


def _target(buf, shape, dev, zero=False):
    """(tensor to write, accumulate flag, value to hand back to autograd)"""
    if buf is not None:
        return buf, 1, None
    t = (torch.zeros if zero else torch.empty)(shape, device=dev, dtype=torch.float32)
    return t, 0, t


# --------------------------------------------------------------------------- #
# Linear
# --------------------------------------------------------------------------- #
class LinearFn(torch.autograd.Function):
    """y = x W^T + b (optional ReLU).  nn.Linear: attention.py:16-18,27; module.py:42-43; encoder.py:27;
    decoder.py:59-60."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _need_cuda(x, w, b)
        M, ldx = _rows(x)
        N, K = w.shape
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        gemm(0, 1, M, N, K, x, ldx, w, K, y, N, bias=b, relu=int(relu))
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = b is not None
        ctx.relu = relu
        ctx.gb = (_gbuf(w), _gbuf(b))
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0).to(dy.dtype)
        M, ldx = _rows(x)
        N, K = w.shape
        dev = dy.device
        dx = dw_ret = db_ret = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dev, dtype=torch.float32)
            gemm(0, 0, M, K, N, dy, N, w, K, dx, K)
        if ctx.needs_input_grad[1]:
            dw, acc, dw_ret = _target(ctx.gb[0], (N, K), dev)
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db, _, db_ret = _target(ctx.gb[1], (N,), dev, zero=True)
            gemm(1, 0, N, K, M, dy, N, x, ldx, dw, K, accumulate=acc, colsum=db)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db, acc, db_ret = _target(ctx.gb[1], (N,), dev)
            call("sbl_colsum_f32", _p(dy), N, _p(db), M, N, acc, _s())
        return dx, dw_ret, db_ret, None


def linear(x, w, b=None, relu=False):
    shp = x.shape
    x2 = x if x.dim() == 2 else x.reshape(-1, shp[-1])
    y = LinearFn.apply(x2, w, b, relu)
    return y if x.dim() == 2 else y.view(*shp[:-1], w.size(0))


# --------------------------------------------------------------------------- #
# dropout / PE / LayerNorm
# --------------------------------------------------------------------------- #
class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _need_cuda(x)
        x = x.contiguous()
        st = dropout_state(x.device)
        ctx.off = st.next_offset()
        ctx.p = p
        ctx.seed = st.seed
        y = torch.empty_like(x)
        call("sbl_dropout", _p(x), _p(y), x.numel(), p, _p(st.seed), ctx.off, _s())
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("sbl_dropout", _p(dy), _p(dx), dy.numel(), ctx.p, _p(ctx.seed), ctx.off, _s())
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p))


class AddPEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe):
        _need_cuda(x, pe)
        B, L, D = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        call("sbl_add_pe", _p(x), _p(pe), _p(y), B, L, D, _s())
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        return dy, None


class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res); encoder.py:54 (res=None), attention.py:58, module.py:51."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        _need_cuda(x, gamma, beta)
        D = x.size(-1)
        x2 = x.contiguous().view(-1, D)
        r2 = None if res is None else res.contiguous().view(-1, D)
        M = x2.size(0)
        y = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        call("sbl_add_layernorm_fwd", _p(x2), _p(r2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, 0.0, None,
             0, _s())
        ctx.save_for_backward(x2, r2, gamma, mean, rstd)
        ctx.shape = x.shape
        ctx.gb = (_gbuf(gamma), _gbuf(beta))
        return y.view(x.shape)

    @staticmethod
    @_bw
    def backward(ctx, dy):
        x2, r2, gamma, mean, rstd = ctx.saved_tensors
        D = x2.size(1)
        M = x2.size(0)
        dy2 = dy.contiguous().view(-1, D)
        dz = torch.empty_like(dy2)
        dg, _, dg_ret = _target(ctx.gb[0], (D,), dy.device, zero=True)
        db, _, db_ret = _target(ctx.gb[1], (D,), dy.device, zero=True)
        call("sbl_add_layernorm_bwd", _p(dy2), _p(x2), _p(r2), _p(gamma), _p(mean), _p(rstd), _p(dz), None, _p(dg), _p(db),
             M, D, 0.0, None, 0, _s())
        dz = dz.view(ctx.shape)
        return dz, (dz if r2 is not None else None), dg_ret, db_ret, None


class RowScaleFn(torch.autograd.Function):
    """x * non_pad_mask (mask: (..., 1) float, one factor per row); encoder.py:86,89."""

    @staticmethod
    def forward(ctx, x, scale):
        _need_cuda(x, scale)
        x = x.contiguous()
        sc = scale.to(torch.float32).contiguous().view(-1)
        D = x.size(-1)
        assert sc.numel() == x.numel() // D
        y = torch.empty_like(x)
        call("sbl_rowscale", _p(x), _p(sc), _p(y), sc.numel(), D, _s())
        ctx.save_for_backward(sc)
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        (sc,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("sbl_rowscale", _p(dy), _p(sc), _p(dx), sc.numel(), dy.size(-1), _s())
        return dx, None


def add_layernorm(x, res, gamma, beta, eps=1e-5):
    return AddLayerNormFn.apply(x, res, gamma, beta, eps)


# --------------------------------------------------------------------------- #
# attention core (used by ScaledDotProductAttention directly)
# --------------------------------------------------------------------------- #
def _mask_args(mask, B, Lq, Lk):
    """mask: None | 'causal' | uint8/bool tensor (B,Lq,Lk) -> (kind, tensor-or-None)"""
    if mask is None:
        return 0, None
    if isinstance(mask, str):
        assert mask == "causal"
        return 1, None
    m = mask
    if m.dim() == 3 and m.size(0) != B:            # head-repeated (n_head*B, Lq, Lk): attention.py:50
        m = m[:B]
    m = m.expand(B, Lq, Lk).to(torch.uint8).contiguous()
    return 2, m


class SDPAFn(torch.autograd.Function):
    """softmax(Q K^T * scale, masked) [dropout] V over (B, L, H*64) views; attention.py:72-83."""

    @staticmethod
    def forward(ctx, q, k, v, H, scale, mask_kind, mask_t, drop_p):
        _need_cuda(q, k, v)
        B, Lq, _ = q.shape
        Lk = k.size(1)
        assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
        assert q.stride(0) == Lq * q.stride(1) and k.stride(0) == Lk * k.stride(1) and v.stride(0) == Lk * v.stride(1)
        o = torch.empty(B, Lq, H * 64, device=q.device, dtype=torch.float32)
        p = torch.empty(H * B, Lq, Lk, device=q.device, dtype=torch.float32)
        seed, off = None, 0
        if drop_p > 0:
            st = dropout_state(q.device)
            seed, off = st.seed, st.next_offset()
        call("sbl_attention_fwd", _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(o), H * 64, _p(p),
             mask_kind, _p(mask_t), B, H, Lq, Lk, scale, drop_p, _p(seed), off, _s())
        ctx.save_for_backward(q, k, v, p, seed)
        ctx.cfg = (H, scale, drop_p, off)
        ctx.mark_non_differentiable(p)
        ctx.set_materialize_grads(False)      # else autograd zero-fills a gradient for p before every backward
        return o, p

    @staticmethod
    @_bw
    def backward(ctx, do, _dp):
        if do is None:
            return (None,) * 8
        q, k, v, p, seed = ctx.saved_tensors
        H, scale, drop_p, off = ctx.cfg
        B, Lq, _ = q.shape
        Lk = k.size(1)
        do = do.contiguous()
        dq = torch.empty(B, Lq, H * 64, device=q.device, dtype=torch.float32)
        dk = torch.empty(B, Lk, H * 64, device=q.device, dtype=torch.float32)
        dv = torch.empty(B, Lk, H * 64, device=q.device, dtype=torch.float32)
        call("sbl_attention_bwd", _p(do), H * 64, _p(q), q.stride(1), _p(k), k.stride(1), _p(v), v.stride(1), _p(p),
             _p(dq), H * 64, _p(dk), H * 64, _p(dv), H * 64, B, H, Lq, Lk, scale, drop_p, _p(seed), off, _s())
        return dq, dk, dv, None, None, None, None, None


# --------------------------------------------------------------------------- #
# fused multi-head attention sub-layer
# --------------------------------------------------------------------------- #
def _adjacent(*ts):
    """True if the tensors are contiguous and laid out back to back in memory (rows of one fused matrix)."""
    p = ts[0].data_ptr()
    for t in ts:
        if not t.is_contiguous() or t.data_ptr() != p:
            return False
        p += t.numel() * t.element_size()
    return True


class KVProjectFn(torch.autograd.Function):
    """[K | V] = x [Wk; Wv]^T + [bk; bv] -> (B*Lk, 2*H*64).  For decoder cross-attention this is hoisted out
    of the 16-step loop (step-invariant: SURVEY 3.2 consequence ii).  Wk/Wv (and bk/bv) must be adjacent rows of
    one fused buffer (MultiHeadAttention keeps them that way)."""

    @staticmethod
    def forward(ctx, x2, wk, bk, wv, bv):
        _need_cuda(x2, wk, wv)
        assert _adjacent(wk, wv) and _adjacent(bk, bv)
        M, ldx = _rows(x2)
        N2, K = 2 * wk.size(0), wk.size(1)
        kv = torch.empty(M, N2, device=x2.device, dtype=torch.float32)
        gemm(0, 1, M, N2, K, x2, ldx, wk, K, kv, N2, bias=bk)
        ctx.save_for_backward(x2, wk, wv)
        gw, gb = (_gbuf(wk), _gbuf(wv)), (_gbuf(bk), _gbuf(bv))
        ctx.gb = (gw[0], gb[0]) if all(g is not None for g in gw + gb) and _adjacent(*gw) and _adjacent(*gb) else (None, None)
        return kv

    @staticmethod
    @_bw
    def backward(ctx, dkv):
        x2, wk, wv = ctx.saved_tensors
        dkv = dkv.contiguous()
        _xs_in(dkv, x2)
        M, ldx = _rows(x2)
        N2, K = 2 * wk.size(0), wk.size(1)
        dev = dkv.device
        dx = torch.empty(M, K, device=dev, dtype=torch.float32)
        _xs_out(dx)
        gemm(0, 0, M, K, N2, dkv, N2, wk, K, dx, K)
        dw, acc, dw_ret = _target(ctx.gb[0], (N2, K), dev)
        db, _, db_ret = _target(ctx.gb[1], (N2,), dev, zero=True)
        wgrad_gemm(N2, K, M, dkv, N2, x2, ldx, dw, K, acc, db)
        _xs_out(dw_ret, db_ret)
        _side_to_main()     # dx joins the other direction's dx in the encoder-output gradient on the main stream
        h = N2 // 2
        if dw_ret is None:
            return dx, None, None, None, None
        return dx, dw_ret[:h], db_ret[:h], dw_ret[h:], db_ret[h:]


class MHAFn(torch.autograd.Function):
    """One whole MultiHeadAttention.forward (attention.py:32-60) as a single tape node:
    projections -> attention core -> fc -> dropout -> LayerNorm(out + residual).

    self_attn=True : q, k, v all come from x through ONE fused (M x 3*H*64) GEMM.
    self_attn=False: q from x, [K|V] given pre-projected (kv, shape (B*Lk, 2*H*64)).
    The output dropout is fused into the LayerNorm kernels, the bias gradients into the weight-gradient GEMMs.
    """

    @staticmethod
    def forward(ctx, x, kv, wq, bq, wk, bk, wv, bv, wfc, bfc, gamma, beta, H, mask_kind, mask_t, drop_p, eps, B, segL):
        """x: (R, D) rows of a ragged batch: segment s has B sequences of length segL[s] (one segment = the plain
        (B, L, D) case).  Returns (y (R, D), p = the segments' (H*B, L, Lk) probability blocks back to back)."""
        _need_cuda(x, wq, wfc)
        x2 = x.contiguous()
        M, D = x2.shape
        assert M == B * sum(segL), (M, B, segL)
        seg_arr, nseg = _segs(segL)
        HD = H * 64
        dev = x.device
        self_attn = kv is None
        if self_attn:
            assert _adjacent(wq, wk, wv) and _adjacent(bq, bk, bv)
            qkv = torch.empty(M, 3 * HD, device=dev, dtype=torch.float32)
            gemm(0, 1, M, 3 * HD, D, x2, D, wq, D, qkv, 3 * HD, bias=bq)
            qp, kp, vp = qkv, qkv[:, HD:], qkv[:, 2 * HD:]
            ldq = ldk = ldv = 3 * HD
            Lk = 0                                   # keys = the segment's own rows
            psize = H * B * sum(l * l for l in segL)
            gw, gb = (_gbuf(wq), _gbuf(wk), _gbuf(wv)), (_gbuf(bq), _gbuf(bk), _gbuf(bv))
            fused_ok = all(g is not None for g in gw + gb) and _adjacent(*gw) and _adjacent(*gb)
            g_qkv = (gw[0], gb[0]) if fused_ok else (None, None)
        else:
            qkv = torch.empty(M, HD, device=dev, dtype=torch.float32)
            gemm(0, 1, M, HD, D, x2, D, wq, D, qkv, HD, bias=bq)
            Lk = kv.size(0) // B
            psize = H * B * sum(segL) * Lk
            qp, kp, vp = qkv, kv, kv[:, HD:]
            ldq, ldk, ldv = HD, 2 * HD, 2 * HD
            g_qkv = (_gbuf(wq), _gbuf(bq))
            if g_qkv[0] is None or g_qkv[1] is None:
                g_qkv = (None, None)
        att = torch.empty(M, HD, device=dev, dtype=torch.float32)
        p = torch.empty(psize, device=dev, dtype=torch.float32)
        seed, off_a, off_o = None, 0, 0
        if drop_p > 0:
            st = dropout_state(dev)
            seed, off_a, off_o = st.seed, st.next_offset(), st.next_offset()
        call("sbl_attention_seg_fwd", _p(qp), ldq, _p(kp), ldk, _p(vp), ldv, _p(att), HD, _p(p), mask_kind, _p(mask_t), B, H,
             seg_arr, nseg, Lk, 1.0 / 8.0, drop_p, _p(seed), off_a, _s())
        o = torch.empty(M, D, device=dev, dtype=torch.float32)
        gemm(0, 1, M, D, HD, att, HD, wfc, HD, o, D, bias=bfc)
        y = torch.empty(M, D, device=dev, dtype=torch.float32)
        mean = torch.empty(M, device=dev, dtype=torch.float32)
        rstd = torch.empty(M, device=dev, dtype=torch.float32)
        call("sbl_add_layernorm_fwd", _p(o), _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, drop_p,
             _p(seed), off_o, _s())
        ctx.save_for_backward(x2, kv, qkv, att, p, o, mean, rstd, wq, wk, wv, wfc, gamma, seed)
        ctx.cfg = (B, tuple(segL), Lk, D, H, drop_p, off_a, off_o, self_attn)
        ctx.gb = (g_qkv, (_gbuf(wfc), _gbuf(bfc)), (_gbuf(gamma), _gbuf(beta)))
        ctx.defer = getattr(_tls, "collector", None)
        ctx.mark_non_differentiable(p)
        ctx.set_materialize_grads(False)      # else autograd zero-fills a gradient for p before every backward
        return y, p

    @staticmethod
    @_bw
    def backward(ctx, dy, _dp):
        if dy is None:
            return (None,) * 19
        x2, kv, qkv, att, p, o, mean, rstd, wq, wk, wv, wfc, gamma, seed = ctx.saved_tensors
        B, segL, Lk, D, H, drop_p, off_a, off_o, self_attn = ctx.cfg
        g_qkv, g_fc, g_ln = ctx.gb
        seg_arr, nseg = _segs(segL)
        M, HD, dev = B * sum(segL), H * 64, dy.device
        _xs_in(dy)
        dy2 = dy.contiguous().view(M, D)
        # LayerNorm(dropout(o) + x) adjoint: dz = grad of the residual x, do = grad of the pre-dropout o
        dz = torch.empty(M, D, device=dev, dtype=torch.float32)
        # a separate pre-dropout gradient buffer is also needed without dropout when the deferred weight-gradient GEMM
        # reads it later while this stream accumulates the input gradient into dz in place
        sep = drop_p > 0 or (ctx.defer is not None and g_fc[0] is not None)
        do = torch.empty(M, D, device=dev, dtype=torch.float32) if sep else None
        dgamma, _, dgamma_ret = _target(g_ln[0], (D,), dev, zero=True)
        dbeta, _, dbeta_ret = _target(g_ln[1], (D,), dev, zero=True)
        call("sbl_add_layernorm_bwd", _p(dy2), _p(o), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(do), _p(dgamma),
             _p(dbeta), M, D, drop_p, _p(seed), off_o, _s())
        if do is None:
            do = dz
        # fc: dW (+ bias grad riding on it), then input gradient
        dwfc, acc, dwfc_ret = _target(g_fc[0], (D, HD), dev)
        dbfc, _, dbfc_ret = _target(g_fc[1], (D,), dev, zero=True)
        wgrad_gemm(D, HD, M, do, D, att, HD, dwfc, HD, acc, dbfc, ctx.defer)
        datt = torch.empty(M, HD, device=dev, dtype=torch.float32)
        gemm(0, 0, M, HD, D, do, D, wfc, HD, datt, HD)
        dx = dz          # residual-branch gradient; the projection's input gradient accumulates on top of it
        if self_attn:
            dqkv = torch.empty(M, 3 * HD, device=dev, dtype=torch.float32)
            ld = 3 * HD
            call("sbl_attention_seg_bwd", _p(datt), HD, _p(qkv), ld, _p(qkv[:, HD:]), ld, _p(qkv[:, 2 * HD:]), ld, _p(p),
                 _p(dqkv), ld, _p(dqkv[:, HD:]), ld, _p(dqkv[:, 2 * HD:]), ld, B, H, seg_arr, nseg, 0, 1.0 / 8.0, drop_p,
                 _p(seed), off_a, _s())
            dw, acc, dw_ret = _target(g_qkv[0], (3 * HD, D), dev)
            db, _, db_ret = _target(g_qkv[1], (3 * HD,), dev, zero=True)
            wgrad_gemm(3 * HD, D, M, dqkv, 3 * HD, x2, D, dw, D, acc, db, ctx.defer)
            gemm(0, 0, M, D, 3 * HD, dqkv, 3 * HD, wq, D, dx, D, accumulate=1)
            if dw_ret is None:
                wret = (None,) * 6
            else:
                wret = (dw_ret[:HD], db_ret[:HD], dw_ret[HD:2 * HD], db_ret[HD:2 * HD], dw_ret[2 * HD:], db_ret[2 * HD:])
            _xs_out(dx)
            return (dx, None) + wret + (dwfc_ret, dbfc_ret, dgamma_ret, dbeta_ret) + (None,) * 7
        dq = torch.empty(M, HD, device=dev, dtype=torch.float32)
        # several segments share the keys/values: their dK/dV contributions are summed inside the call
        dkv = torch.empty(B * Lk, 2 * HD, device=dev, dtype=torch.float32)
        call("sbl_attention_seg_bwd", _p(datt), HD, _p(qkv), HD, _p(kv), 2 * HD, _p(kv[:, HD:]), 2 * HD, _p(p), _p(dq), HD,
             _p(dkv), 2 * HD, _p(dkv[:, HD:]), 2 * HD, B, H, seg_arr, nseg, Lk, 1.0 / 8.0, drop_p, _p(seed), off_a, _s())
        dwq, acc, dwq_ret = _target(g_qkv[0], (HD, D), dev)
        dbq, _, dbq_ret = _target(g_qkv[1], (HD,), dev, zero=True)
        wgrad_gemm(HD, D, M, dq, HD, x2, D, dwq, D, acc, dbq, ctx.defer)
        gemm(0, 0, M, D, HD, dq, HD, wq, D, dx, D, accumulate=1)
        _xs_out(dx, dkv)
        return (dx, dkv, dwq_ret, dbq_ret, None, None, None, None, dwfc_ret, dbfc_ret, dgamma_ret, dbeta_ret) + (None,) * 7


# tests only: callable(w1, h) handed every feed-forward's post-ReLU hidden activation (tests/test_hip_parity.py compares the
# ReLU masks of this path and of the CPU oracle, to tell a flipped mask bit from an arithmetic error)
_ffn_probe = None


class FFNFn(torch.autograd.Function):
    """PositionwiseFeedForward.forward (module.py:47-52) as one tape node:
    LayerNorm(dropout(relu(x W1^T + b1) W2^T + b2) + x); dropout fused into the LayerNorm kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, drop_p, eps):
        _need_cuda(x, w1, w2)
        shp = x.shape
        D = shp[-1]
        x2 = x.contiguous().view(-1, D)
        M, F_, dev = x2.size(0), w1.size(0), x.device
        h = torch.empty(M, F_, device=dev, dtype=torch.float32)
        gemm(0, 1, M, F_, D, x2, D, w1, D, h, F_, bias=b1, relu=1)
        if _ffn_probe is not None:
            _ffn_probe(w1, h)
        o = torch.empty(M, D, device=dev, dtype=torch.float32)
        gemm(0, 1, M, D, F_, h, F_, w2, F_, o, D, bias=b2)
        seed, off = None, 0
        if drop_p > 0:
            st = dropout_state(dev)
            seed, off = st.seed, st.next_offset()
        y = torch.empty(M, D, device=dev, dtype=torch.float32)
        mean = torch.empty(M, device=dev, dtype=torch.float32)
        rstd = torch.empty(M, device=dev, dtype=torch.float32)
        call("sbl_add_layernorm_fwd", _p(o), _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, drop_p,
             _p(seed), off, _s())
        ctx.save_for_backward(x2, h, o, mean, rstd, w1, w2, gamma, seed)
        ctx.cfg = (shp, drop_p, off)
        ctx.gb = ((_gbuf(w1), _gbuf(b1)), (_gbuf(w2), _gbuf(b2)), (_gbuf(gamma), _gbuf(beta)))
        ctx.defer = getattr(_tls, "collector", None)
        return y.view(shp)

    @staticmethod
    @_bw
    def backward(ctx, dy):
        x2, h, o, mean, rstd, w1, w2, gamma, seed = ctx.saved_tensors
        shp, drop_p, off = ctx.cfg
        g1, g2, g_ln = ctx.gb
        M, D = x2.shape
        F_, dev = w1.size(0), dy.device
        _xs_in(dy)
        dy2 = dy.contiguous().view(M, D)
        dz = torch.empty(M, D, device=dev, dtype=torch.float32)
        # (same as in MHAFn.backward)
        sep = drop_p > 0 or (ctx.defer is not None and g2[0] is not None)
        do = torch.empty(M, D, device=dev, dtype=torch.float32) if sep else None
        dgamma, _, dgamma_ret = _target(g_ln[0], (D,), dev, zero=True)
        dbeta, _, dbeta_ret = _target(g_ln[1], (D,), dev, zero=True)
        call("sbl_add_layernorm_bwd", _p(dy2), _p(o), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(do), _p(dgamma),
             _p(dbeta), M, D, drop_p, _p(seed), off, _s())
        if do is None:
            do = dz
        dw2, acc, dw2_ret = _target(g2[0], (D, F_), dev)
        db2, _, db2_ret = _target(g2[1], (D,), dev, zero=True)
        wgrad_gemm(D, F_, M, do, D, h, F_, dw2, F_, acc, db2, ctx.defer)
        dh = torch.empty(M, F_, device=dev, dtype=torch.float32)
        gemm(0, 0, M, F_, D, do, D, w2, F_, dh, F_, mask=h, ldm=F_)      # ReLU adjoint fused in the epilogue
        dw1, acc, dw1_ret = _target(g1[0], (F_, D), dev)
        db1, _, db1_ret = _target(g1[1], (F_,), dev, zero=True)
        wgrad_gemm(F_, D, M, dh, F_, x2, D, dw1, D, acc, db1, ctx.defer)
        dx = dz
        gemm(0, 0, M, D, F_, dh, F_, w1, D, dx, D, accumulate=1)
        _xs_out(dx)
        return dx.view(shp), dw1_ret, db1_ret, dw2_ret, db2_ret, dgamma_ret, dbeta_ret, None, None


# --------------------------------------------------------------------------- #
# decoder pieces
# --------------------------------------------------------------------------- #
class EmbedPEFn(torch.autograd.Function):
    """emb[tok] + pe[:L] for every segment (decoder step) of a run; decoder.py:116-120 (x_logit_scale = 1).
    tok: (B, 17) device token buffer; segment s embeds its first segL[s] columns.  Returns (B*sum(segL), D)."""

    @staticmethod
    def forward(ctx, tok, B, segL, emb, pe):
        _need_cuda(tok, emb, pe)
        V, D = emb.shape
        seg_arr, nseg = _segs(segL)
        out = torch.empty(B * sum(segL), D, device=emb.device, dtype=torch.float32)
        call("sbl_embed_pe_seg_fwd", _p(tok), tok.stride(0), _p(emb), _p(pe), _p(out), B, seg_arr, nseg, D, V, _s())
        # tokens are written in place by later steps, but positions inside an already embedded prefix never change
        ctx.tok, ctx.B, ctx.segL, ctx.shape = tok, B, tuple(segL), (V, D)
        ctx.gb = _gbuf(emb)
        return out

    @staticmethod
    @_bw
    def backward(ctx, dy):
        V, D = ctx.shape
        dy = dy.contiguous()
        seg_arr, nseg = _segs(ctx.segL)
        demb, _, demb_ret = _target(ctx.gb, (V, D), dy.device, zero=True)
        call("sbl_embed_seg_bwd", _p(ctx.tok), ctx.tok.stride(0), _p(dy), _p(demb), ctx.B, seg_arr, nseg, D, V, _s())
        return None, None, None, demb_ret, None


class FusionFn(torch.autograd.Function):
    """A' = A + flip(B), B' = 2B + flip(A) along each sequence's own prefix; decoder.py:132-143,160-164
    (closed form).  a, b: (B*sum(segL), D) rows, or (B, L, D) with segL=None."""

    @staticmethod
    def forward(ctx, a, b, B=None, segL=None):
        _need_cuda(a, b)
        a, b = a.contiguous(), b.contiguous()
        if segL is None:
            B, segL = a.size(0), (a.size(1),)
        D = a.size(-1)
        seg_arr, nseg = _segs(segL)
        a2, b2 = torch.empty_like(a), torch.empty_like(b)
        call("sbl_fusion_seg_fwd", _p(a), _p(b), _p(a2), _p(b2), B, seg_arr, nseg, D, _s())
        ctx.cfg = (B, tuple(segL), D)
        return a2, b2

    @staticmethod
    @_bw
    def backward(ctx, da2, db2):
        B, segL, D = ctx.cfg
        da2, db2 = da2.contiguous(), db2.contiguous()
        seg_arr, nseg = _segs(segL)
        da, db = torch.empty_like(da2), torch.empty_like(db2)
        call("sbl_fusion_seg_bwd", _p(da2), _p(db2), _p(da), _p(db), B, seg_arr, nseg, D, _s())
        return da, db, None, None


class GatherLastFn(torch.autograd.Function):
    """The last position of every sequence of a ragged batch, (nseg*B, D): the rows the heads read, decoder.py:166-167."""

    @staticmethod
    def forward(ctx, x, B, segL):
        _need_cuda(x)
        x = x.contiguous()
        D = x.size(-1)
        seg_arr, nseg = _segs(segL)
        out = torch.empty(nseg * B, D, device=x.device, dtype=torch.float32)
        call("sbl_gather_last_fwd", _p(x), _p(out), B, seg_arr, nseg, D, _s())
        ctx.cfg = (B, tuple(segL), D, x.shape)
        return out

    @staticmethod
    @_bw
    def backward(ctx, dy):
        B, segL, D, shp = ctx.cfg
        dy = dy.contiguous()
        seg_arr, nseg = _segs(segL)
        dx = torch.empty(shp, device=dy.device, dtype=torch.float32)
        call("sbl_gather_last_bwd", _p(dy), _p(dx), B, seg_arr, nseg, D, _s())
        return dx, None, None


def argmax_select(pred, gold, ys, step, use_argmax, coins_dev=None):
    """ys[:, step+1] = argmax(pred) if coin else gold[:, step]; decoder.py:173-186, on the device."""
    if pred is None:                       # pure teacher token (no logits needed)
        assert not use_argmax and coins_dev is None and gold is not None
        B, V, ldp = ys.size(0), 1, 1
    else:
        (B, V), ldp = pred.shape, pred.stride(0)
    call("sbl_argmax_select", _p(pred), ldp, _p(gold), 0 if gold is None else gold.stride(0), _p(ys),
         ys.stride(0), step, int(use_argmax), _p(coins_dev), B, V, _s())


class SmoothedCEFn(torch.autograd.Function):
    """cal_loss (loss.py:27-52): returns (mean loss over gold != ignore, stats[3] = (sum, n_valid, n_correct))."""

    @staticmethod
    def forward(ctx, pred, gold, eps, ignore_id):
        _need_cuda(pred, gold)
        pred = pred.contiguous()
        gold = gold.contiguous()
        R, C = pred.shape
        out3 = torch.empty(3, device=pred.device, dtype=torch.float32)
        call("sbl_smoothed_ce_fwd", _p(pred), _p(gold), _p(out3), R, C, eps, ignore_id, _s())
        ctx.save_for_backward(pred, gold, out3)
        ctx.cfg = (eps, ignore_id)
        ctx.mark_non_differentiable(out3)
        ctx.set_materialize_grads(False)
        return out3[0] / out3[1], out3

    @staticmethod
    @_bw
    def backward(ctx, gloss, _g3):
        if gloss is None:
            return None, None, None, None
        pred, gold, out3 = ctx.saved_tensors
        eps, ignore_id = ctx.cfg
        R, C = pred.shape
        gs = gloss.reshape(1).to(torch.float32).contiguous()
        dpred = torch.empty_like(pred)
        call("sbl_smoothed_ce_bwd", _p(pred), _p(gold), _p(out3), _p(gs), _p(dpred), R, C, eps, ignore_id, _s())
        return dpred, None, None, None


# --------------------------------------------------------------------------- #
# visual frontend
# --------------------------------------------------------------------------- #
# Packed convolution weights.  The state dict keeps OIHW; the kernels read OHWI (forward / weight gradient) and
# [Cin][kh][kw][Cout] (input gradient).  Weights change once per optimizer step, not once per forward, so the two packed
# images of each of the 19 trunk convolutions are kept and re-made only when the weight changed:
#   * torch in-place updates (torch.optim, load_state_dict, copy_) bump `w._version` - or the flat buffer's version when the
#     parameter is a view of a dp.FlatModel - and the next forward re-packs;
#   * writers that bypass torch (FusedAdam's raw kernel on the flat buffer) call refresh_packed_weights(), which re-packs
#     every cached image IN PLACE right away, so hipGraphs captured with the cached images stay valid across optimizer steps.
# A captured graph that is replayed between FOREIGN in-place updates must call refresh_packed_weights() itself.
PACK_CACHE = True
_packed = {}          # id(weight) -> entry
_pack_epoch = [0]


def _pack_key(w):
    fm = getattr(w, "_sbl_flat", None)
    return (w.data_ptr(), w._version, fm.flat_param._version if fm is not None else -1, _pack_epoch[0])


def _pack_entry_alive(e):
    return e["ref"]() is not None


def packed_conv_weight(w, need_dg, stats):
    """(w_ohwi, w_dg) for an OIHW convolution weight; `stats` (fp64, 2*Cout, or None) is zero-filled for the caller either
    by the pack launch (as before) or, on a cache hit, here."""
    import weakref
    Cout, Cin, KH, KW = w.shape
    dev = w.device
    e = _packed.get(id(w)) if PACK_CACHE else None
    if e is not None and (e["ref"]() is not w or e["shape"] != tuple(w.shape)):
        e = None
    if e is not None and e["key"] == _pack_key(w) and (e["dg"] is not None or not need_dg):
        if stats is not None:
            stats.zero_()
        return e["ohwi"], e["dg"]
    w_ohwi = e["ohwi"] if e is not None else torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
    w_dg = e["dg"] if (e is not None and e["dg"] is not None) else (torch.empty(Cin, KH, KW, Cout, device=dev, dtype=torch.float32) if need_dg else None)
    call("sbl_conv_weight_pack", _p(w.detach().contiguous()), _p(w_ohwi), _p(w_dg), Cout, Cin, KH, KW, _p(stats),
         stats.numel() if stats is not None else 0, _s())
    if PACK_CACHE:
        _packed[id(w)] = {"ref": weakref.ref(w), "shape": tuple(w.shape), "key": _pack_key(w), "ohwi": w_ohwi, "dg": w_dg}
        if len(_packed) > 256:
            for k in [k for k, v in _packed.items() if not _pack_entry_alive(v)]:
                del _packed[k]
    return w_ohwi, w_dg


def refresh_packed_weights():
    """Re-pack every cached image in place from the current weights (current stream).  FusedAdam.step calls it; callers
    that update parameters by other raw kernels, or replay captured graphs between foreign in-place updates, must too."""
    _pack_epoch[0] += 1
    for k, e in list(_packed.items()):
        w = e["ref"]()
        if w is None or not w.is_cuda:
            del _packed[k]
            continue
        Cout, Cin, KH, KW = w.shape
        call("sbl_conv_weight_pack", _p(w.detach().contiguous()), _p(e["ohwi"]), _p(e["dg"]), Cout, Cin, KH, KW, None, 0, _s())
        e["key"] = _pack_key(w)


class StemFn(torch.autograd.Function):
    """frontend3D (video_frontend.py:99-104) on (N,T,H,W) clips -> pooled NHWC (N*T, H/4, W/4, 64)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, training, momentum, eps, nbt=None):
        """nbt: the BatchNorm's num_batches_tracked (int64 scalar) or None; incremented by the finalize kernel."""
        _need_cuda(x, w, gamma, beta)
        x = x.contiguous()
        N, T, H, W = x.shape
        dev = x.device
        if training and any(ctx.needs_input_grad):
            zero_pool_reset(dev)          # the step starts here: one memset for every zero-initialised buffer of its backward
        Ho, Wo = H // 2, W // 2
        w2 = w.contiguous().view(64, 245)
        conv = torch.empty(N * T, Ho, Wo, 64, device=dev, dtype=torch.float32)
        stats = torch.empty(128, device=dev, dtype=torch.float64)
        call("sbl_stem_conv_fwd", _p(x), _p(w2), _p(conv), _p(stats), N, T, H, W, _s())
        mean = torch.empty(64, device=dev, dtype=torch.float32)
        invstd = torch.empty(64, device=dev, dtype=torch.float32)
        if training:
            call("sbl_bn_finalize", _p(stats), N * T * Ho * Wo, _p(running_mean), _p(running_var), momentum, eps, _p(mean),
                 _p(invstd), 64, _p(nbt), _s())
        else:
            call("sbl_bn_eval_stats", _p(running_mean), _p(running_var), eps, _p(mean), _p(invstd), 64, _s())
        pooled = torch.empty(N * T, Ho // 2, Wo // 2, 64, device=dev, dtype=torch.float32)
        argmax = torch.empty(N * T, Ho // 2, Wo // 2, 64, device=dev, dtype=torch.uint8)
        call("sbl_stem_bn_relu_pool_fwd", _p(conv), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(pooled), _p(argmax),
             N * T, Ho, Wo, _s())
        ctx.save_for_backward(x, conv, argmax, mean, invstd, gamma, beta)
        ctx.training = training
        return pooled

    @staticmethod
    @_bw
    def backward(ctx, dpooled):
        x, conv, argmax, mean, invstd, gamma, beta = ctx.saved_tensors
        if not ctx.training:
            raise _lib.SblHipError("stem backward is implemented for training-mode BatchNorm (batch statistics) only")
        N, T, H, W = x.shape
        dev = x.device
        dpooled = dpooled.contiguous()
        sums = torch.empty(128, device=dev, dtype=torch.float64)
        call("sbl_stem_bwd_reduce", _p(conv), _p(dpooled), _p(argmax), _p(mean), _p(invstd), _p(gamma), _p(beta),
             _p(sums), N * T, H // 2, W // 2, _s())
        dw = torch.empty(64, 245, device=dev, dtype=torch.float32)
        dgamma = torch.empty(64, device=dev, dtype=torch.float32)
        dbeta = torch.empty(64, device=dev, dtype=torch.float32)
        call("sbl_stem_wgrad", _p(x), _p(conv), _p(dpooled), _p(argmax), _p(mean), _p(invstd), _p(gamma), _p(beta),
             _p(sums), _p(dw), _p(dgamma), _p(dbeta), N, T, H, W, _s())
        return None, dw.view(64, 1, 5, 7, 7), dgamma, dbeta, None, None, None, None, None, None


class ConvBNFn(torch.autograd.Function):
    """conv (3x3 pad 1 | 1x1 pad 0, bias-free) -> BatchNorm2d -> [+ residual] -> [ReLU] on NHWC activations;
    video_frontend.py:28-41,69-71.  The BN batch statistics are reduced in the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, res, relu, stride, training, momentum, eps, box_out=None,
                box_in=None, nbt=None, ctl=None):
        """box_out / box_in (dicts or None) link conv1 -> bn1 -> relu to the conv2 that consumes it (BasicBlock,
        video_frontend.py:31-35): this node publishes (pre-BN output, mean, invstd) in box_out; the consumer, given the same
        dict as box_in, computes bn1's backward reduction in the epilogue of its input-gradient convolution
        (sbl_conv2d_dgrad_bnstats) and leaves the sums there, so this node's backward skips its reduction pass."""
        _need_cuda(x, w, gamma, beta)
        x = x.contiguous()
        NIMG, H, W, Cin = x.shape
        Cout, _, KH, KW = w.shape
        pad = 1 if KH == 3 else 0
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        dev = x.device
        # packed weight images (OHWI; and [Cin][kh][kw][Cout] for the input gradient, kept for backward): cached across
        # steps, see packed_conv_weight.  The BN statistics the convolution's epilogue accumulates into come zero-filled from
        # the step's pooled memset (or are zeroed by the pack launch / a fill when the pool is not armed).
        # (Measured and not kept in round 2: all 19 packs on the side stream while the stem runs - same-box A/B 32.94 vs
        # 32.72 ms, the fork/join and the contention with the stem cost more than the 10 us per convolution they take off.)
        stats = None
        pooled_stats = False
        if training:
            stats = zero_pool_take(dev, (2 * Cout,), torch.float64)
            pooled_stats = stats is not None
            if not pooled_stats:
                stats = torch.empty(2 * Cout, device=dev, dtype=torch.float64)
        w_ohwi, w_dg = packed_conv_weight(w, training and ctx.needs_input_grad[0], None if pooled_stats else stats)
        conv = torch.empty(NIMG, Ho, Wo, Cout, device=dev, dtype=torch.float32)
        mean = torch.empty(Cout, device=dev, dtype=torch.float32)
        invstd = torch.empty(Cout, device=dev, dtype=torch.float32)
        y = torch.empty_like(conv)
        r = None if res is None else res.contiguous()
        if training:
            call("sbl_conv2d_fwd", _p(x), _p(w_ohwi), _p(conv), _p(stats), 1, NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                 _workspace().data_ptr(), WS_BYTES, _s())
            # the BatchNorm "finalize" (mean / invstd / running statistics / num_batches_tracked) rides on the apply launch
            call("sbl_bn_apply_fwd_stats", _p(conv), _p(r), _p(stats), NIMG * Ho * Wo, _p(running_mean), _p(running_var), momentum, eps,
                 _p(gamma), _p(beta), _p(y), _p(mean), _p(invstd), _p(nbt), NIMG * Ho * Wo, Cout, int(relu), _s())
        else:
            call("sbl_conv2d_fwd", _p(x), _p(w_ohwi), _p(conv), None, 0, NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                 _workspace().data_ptr(), WS_BYTES, _s())
            call("sbl_bn_eval_stats", _p(running_mean), _p(running_var), eps, _p(mean), _p(invstd), Cout, _s())
            call("sbl_bn_apply_fwd", _p(conv), _p(r), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(y), NIMG * Ho * Wo, Cout,
                 int(relu), _s())
        ctx.save_for_backward(x, w, conv, y if relu else None, mean, invstd, gamma, w_dg)
        ctx.cfg = (relu, stride, pad, training, res is not None)
        ctx.gb_bn = (_gbuf(gamma), _gbuf(beta))
        ctx.box_out = ctx.box_in = None
        # ctl (BasicBlock, video_frontend.py:28-41) = {"role": "conv1" | "ds" | "conv2", "link": dict shared by the block's
        # three nodes, "prev": the dict the PREVIOUS block published on its output tensor, "pub": the dict this block
        # publishes}.  It lets backward (a) fold the residual branch's gradient into conv1's input-gradient epilogue instead
        # of an autograd add, and (b) reduce the previous block's bn2 (+ downsample BN) backward sums in that same epilogue.
        ctx.ctl = ctl if training else None
        if ctx.ctl is not None:
            if ctl["role"] == "conv2":
                # (only the address of the activation is kept: the dict hangs on y itself (`y._sbl_pub`), a tensor in it
                # would be a reference cycle that only the cyclic GC frees - ~1 GB of trunk activations per step)
                ctl["pub"].update(conv=conv, mean=mean, invstd=invstd, act=y.data_ptr())
            elif ctl["role"] == "ds":
                ctl["pub"].update(conv2=conv, mean2=mean, invstd2=invstd)
        if training:      # (grad mode is off inside Function.forward; backward only runs if a tape exists)
            if box_out is not None and relu and res is None:
                box_out.update(conv=conv, mean=mean, invstd=invstd, act=y.data_ptr())
                ctx.box_out = box_out
            if box_in is not None and stride == 1 and box_in.get("act") == x.data_ptr():
                ctx.box_in = box_in
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        x, w, conv, y, mean, invstd, gamma, w_dg = ctx.saved_tensors
        relu, stride, pad, training, has_res = ctx.cfg
        if not training:
            raise _lib.SblHipError("ConvBN backward is implemented for training-mode BatchNorm only")
        NIMG, H, W, Cin = x.shape
        Cout, _, KH, KW = w.shape
        dev = x.device
        dy = dy.contiguous()
        rows = conv.numel() // Cout
        box = ctx.box_out
        ctl = ctx.ctl
        role = ctl["role"] if ctl is not None else None
        pub = ctl["pub"] if role in ("conv2", "ds") else None
        if box is not None and box.get("sums") is not None and box.get("dx_ptr") == dy.data_ptr():
            sums = box.pop("sums")           # reduced in the epilogue of the consumer's input-gradient convolution
            box.clear()
        elif role == "conv2" and pub.get("sums") is not None and pub.get("dx_ptr") == dy.data_ptr():
            # the NEXT block's conv1 input-gradient epilogue reduced them (dy is that convolution's output, residual included)
            full = pub.pop("sums")
            sums = full[:2 * Cout]
            if full.numel() == 4 * Cout:
                ctl["link"]["ds_sums"] = full[2 * Cout:]     # the downsample BatchNorm's pair: same g, its own xhat
            pub.clear()                                       # consumed: drop the published tensors now
        elif role == "ds" and ctl["link"].get("ds_sums") is not None:
            sums = ctl["link"].pop("ds_sums")
        else:
            sums = torch.empty(2 * Cout, device=dev, dtype=torch.float64)
            call("sbl_bn_bwd_reduce", _p(dy), _p(y), _p(conv), _p(mean), _p(invstd), _p(sums), rows, Cout, int(relu),
                 _workspace().data_ptr(), WS_BYTES, _s())
        dconv = torch.empty_like(conv)
        dres = torch.empty_like(conv) if has_res else None
        if ctx.gb_bn[0] is not None and ctx.gb_bn[1] is not None:      # persistent gradient buffers: += in the kernel
            call("sbl_bn_bwd_apply", _p(dy), _p(y), _p(conv), _p(mean), _p(invstd), _p(gamma), _p(sums), _p(dconv), _p(dres),
                 _p(ctx.gb_bn[0]), _p(ctx.gb_bn[1]), rows, Cout, int(relu), 1, _s())
            dgamma = dbeta = None
        else:
            dgamma = torch.empty(Cout, device=dev, dtype=torch.float32)
            dbeta = torch.empty(Cout, device=dev, dtype=torch.float32)
            call("sbl_bn_bwd_apply", _p(dy), _p(y), _p(conv), _p(mean), _p(invstd), _p(gamma), _p(sums), _p(dconv), _p(dres),
                 _p(dgamma), _p(dbeta), rows, Cout, int(relu), 0, _s())
        dres_ret = dres
        if role == "conv2" and ctl["link"].get("identity"):
            ctl["link"]["dres"] = dres        # conv1's input-gradient epilogue adds it (conv1's backward runs after this one)
            dres_ret = None
        dx = None
        if ctx.needs_input_grad[0] and role == "ds" and not ctl["link"].get("main_done"):
            # 1x1 / stride-2: the gradient lives on the even/even pixels; hand the compact form to conv1's epilogue (this node
            # was created after conv1, so the engine runs it first) instead of a zero-filled full-size tensor + autograd add
            if w_dg is None:
                w_ohwi = torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
                w_dg = torch.empty(Cin, KH, KW, Cout, device=dev, dtype=torch.float32)
                call("sbl_conv_weight_pack", _p(w.contiguous()), _p(w_ohwi), _p(w_dg), Cout, Cin, KH, KW, None, 0, _s())
            dxc = torch.empty(NIMG, (H + 1) // 2, (W + 1) // 2, Cin, device=dev, dtype=torch.float32)
            call("sbl_conv1x1s2_dgrad_compact", _p(dconv), _p(w_dg), _p(dxc), NIMG, H, W, Cin, Cout, _workspace().data_ptr(), WS_BYTES, _s())
            ctl["link"]["dx_ds"] = dxc
        elif ctx.needs_input_grad[0]:
            if w_dg is None:
                w_ohwi = torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
                w_dg = torch.empty(Cin, KH, KW, Cout, device=dev, dtype=torch.float32)
                call("sbl_conv_weight_pack", _p(w.contiguous()), _p(w_ohwi), _p(w_dg), Cout, Cin, KH, KW, None, 0, _s())
            dx = torch.empty_like(x)
            bi = ctx.box_in
            if role == "conv1":
                link = ctl["link"]
                addend = link.pop("dres", None) if link.get("identity") else link.pop("dx_ds", None)
                link["main_done"] = True
                prev = ctl.get("prev")
                fuse = prev is not None and addend is not None and prev.get("act") == x.data_ptr()
                nsums = None
                if fuse:
                    two = prev.get("conv2") is not None
                    nsums = zero_pool_take(dev, ((4 if two else 2) * Cin,), torch.float64)
                    pooled = nsums is not None
                    if not pooled:
                        nsums = torch.empty((4 if two else 2) * Cin, device=dev, dtype=torch.float64)
                    call("sbl_conv2d_dgrad_fused", _p(dconv), _p(w_dg), _p(dx), NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                         _workspace().data_ptr(), WS_BYTES, _p(addend), _p(x), _p(prev["conv"]), _p(prev["mean"]), _p(prev["invstd"]),
                         _p(prev.get("conv2")), _p(prev.get("mean2")), _p(prev.get("invstd2")), _p(nsums), int(pooled), _s())
                    prev["sums"], prev["dx_ptr"] = nsums, dx.data_ptr()
                else:
                    call("sbl_conv2d_dgrad_fused", _p(dconv), _p(w_dg), _p(dx), NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                         _workspace().data_ptr(), WS_BYTES, _p(addend), None, None, None, None, None, None, None, None, 0, _s())
            elif bi is not None and bi.get("conv") is not None:
                nsums = zero_pool_take(dev, (2 * Cin,), torch.float64)
                pooled = nsums is not None
                if not pooled:
                    nsums = torch.empty(2 * Cin, device=dev, dtype=torch.float64)
                call("sbl_conv2d_dgrad_bnstats", _p(dconv), _p(w_dg), _p(dx), NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                     _workspace().data_ptr(), WS_BYTES, _p(x), _p(bi["conv"]), _p(bi["mean"]), _p(bi["invstd"]), _p(nsums), int(pooled), _s())
                bi["sums"], bi["dx_ptr"] = nsums, dx.data_ptr()
            else:
                call("sbl_conv2d_dgrad", _p(dconv), _p(w_dg), _p(dx), NIMG, H, W, Cin, Cout, KH, KW, stride, pad,
                     _workspace().data_ptr(), WS_BYTES, _s())
        gw = _gbuf(w)
        if gw is not None:
            # the weight gradient is off backward's dependency chain: with a persistent gradient buffer it is issued
            # on the second stream, where its workgroups fill the CUs that the chain's kernels (tile-count
            # quantisation: 522 workgroups on 256 CUs) leave idle; joined by an end-of-backward engine callback
            cur = torch.cuda.current_stream()
            side = side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dw_ohwi = zero_pool_take(dev, (Cout, KH, KW, Cin), torch.float32)
                pooled = dw_ohwi is not None
                if not pooled:
                    dw_ohwi = torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
                call("sbl_conv2d_wgrad", _p(x), _p(dconv), _p(dw_ohwi), NIMG, H, W, Cin, Cout, KH, KW, stride, pad, int(pooled), _s())
                call("sbl_conv_wgrad_unpack", _p(dw_ohwi), _p(gw), Cout, Cin, KH, KW, 1, _s())
            x.record_stream(side)
            dconv.record_stream(side)
            _arm_side_join()
            return dx, None, dgamma, dbeta, None, None, dres_ret, None, None, None, None, None, None, None, None, None
        dw_ohwi = torch.empty(Cout, KH, KW, Cin, device=dev, dtype=torch.float32)
        call("sbl_conv2d_wgrad", _p(x), _p(dconv), _p(dw_ohwi), NIMG, H, W, Cin, Cout, KH, KW, stride, pad, 0, _s())
        if gw is not None:
            call("sbl_conv_wgrad_unpack", _p(dw_ohwi), _p(gw), Cout, Cin, KH, KW, 1, _s())
            return dx, None, dgamma, dbeta, None, None, dres_ret, None, None, None, None, None, None, None, None, None
        dw = torch.empty_like(w)
        call("sbl_conv_wgrad_unpack", _p(dw_ohwi), _p(dw), Cout, Cin, KH, KW, 0, _s())
        return dx, dw, dgamma, dbeta, None, None, dres_ret, None, None, None, None, None, None, None, None, None


class AvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + view on NHWC; video_frontend.py:87-88."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        NIMG, H, W, C = x.shape
        y = torch.empty(NIMG, C, device=x.device, dtype=torch.float32)
        call("sbl_avgpool_fwd", _p(x), _p(y), NIMG, H * W, C, _s())
        ctx.shape = x.shape
        return y

    @staticmethod
    @_bw
    def backward(ctx, dy):
        NIMG, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        call("sbl_avgpool_bwd", _p(dy), _p(dx), NIMG, H * W, C, _s())
        return dx


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """Fused Adam over flat fp32 buffers (SBL/train.py:75; optimizer.py:18-27)."""
    _need_cuda(p, g, m, v)
    call("sbl_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _s())


_LUT = {}


def preprocess_clips(frames_u8, y1, x1, flip, src_frame, Tout=30, crop=(88, 88), mean=0.413621, std=0.1700239):
    """Device input pipeline (SBL/data_gen.py:276-296 + cvtransforms.py): uint8 (N,Tin,Hin,Win) grayscale frames ->
    normalised, cropped, flipped, frame-mapped, zero-padded fp32 clips (N,Tout,Hc,Wc) in one kernel.  y1/x1/flip:
    int32 (N,) device tensors; src_frame: int32 (N,Tout), -1 = zero frame."""
    import numpy as np
    _need_cuda(frames_u8, y1, x1, flip, src_frame)
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    N, Tin, Hin, Win = frames_u8.shape
    key = (frames_u8.device.index, mean, std)
    lut = _LUT.get(key)
    if lut is None:   # float32((v/255. - mean)/std) in double, exactly the reference's numpy arithmetic
        lut = _LUT[key] = torch.from_numpy(((np.arange(256, dtype=np.float64) / 255. - mean) / std).astype(np.float32)).to(frames_u8.device)
    out = torch.empty(N, Tout, crop[0], crop[1], device=frames_u8.device, dtype=torch.float32)
    call("sbl_preprocess_clips", _p(frames_u8), _p(out), _p(lut), _p(y1.contiguous()), _p(x1.contiguous()), _p(flip.contiguous()),
         _p(src_frame.contiguous()), N, Tin, Hin, Win, Tout, crop[0], crop[1], _s())
    return out
