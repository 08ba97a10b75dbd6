"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise FlatModel (flat parameter / gradient
buffers, fused-QKV adjacency preserved) and GradientExchange (segment-wise all-reduce average, with and without
the backward-overlap hooks).  The model's HIP forward is not run here (no GPU): gradients are injected directly,
which is exactly what the exchange sees after a real backward."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_model():
    from sbl_for_multilingual_lip_reading_amd.transformer.decoder import Decoder
    from sbl_for_multilingual_lip_reading_amd.transformer.encoder import Encoder
    from sbl_for_multilingual_lip_reading_amd.transformer.transformer import Transformer
    torch.manual_seed(0)
    return Transformer(Encoder(512, 1, 8, 64, 64, 512, 2048), Decoder(0, 1, 58, 512, 1, 8, 64, 64, 512, 2048), None)


def _worker(rank, world, port, overlap, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbl_for_multilingual_lip_reading_amd import dp
    m = _small_model()
    if rank == 1:                                   # replicas start different: broadcast must fix that
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    flat = dp.FlatModel(m)
    # flattening must not change values, and must keep q/k/v weights adjacent
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
        assert p.data_ptr() % 16 == 0
    mha = m.encoder.layer_stack[0].slf_attn
    assert mha.w_ks.weight.data_ptr() == mha.w_qs.weight.data_ptr() + mha.w_qs.weight.numel() * 4
    assert mha.w_vs.bias.data_ptr() == mha.w_ks.bias.data_ptr() + mha.w_ks.bias.numel() * 4
    mha._fuse()                                      # no-op now
    assert mha.w_qs.weight.data_ptr() >= flat.flat_param.data_ptr()
    dp.broadcast_parameters(flat)
    # small buckets so that every segment goes out as several all-reduces; the "sum" variant leaves the 1/world to the
    # optimizer (FusedAdam(grad_scale=1/world))
    ex = dp.GradientExchange(flat, world, overlap=overlap, average=(overlap != "sum"), bucket_bytes=1 << 20)
    flat.zero_grad()
    # "backward": every parameter's gradient = (rank+1) * its index pattern, accumulated in place like AccumulateGrad
    for i, (n, p) in enumerate(m.named_parameters()):
        p.grad.add_(float(rank + 1) * (1.0 + (i % 7)))
    if overlap:                                      # fire the segment launches in backward order, like the hooks do
        assert len(ex._hooks) == 5                   # encoder out, frontend out, inputs of ResNet stages 4, 3, 2
        for seg in dp.FlatModel.SEGMENTS[:-1]:
            ex.launch(seg)
        try:                                         # a second backward before finish() is refused, not swallowed
            ex.launch("decoder.")
            raise AssertionError("second launch of an exchanged segment must raise")
        except RuntimeError as e:
            assert "finish()" in str(e)
    ex.finish()
    # reverse-autograd order, <= 1 MB each, every element exactly once
    order = [s for s, _ in ex.launches]
    assert [s for i, s in enumerate(order) if i == 0 or order[i - 1] != s] == list(dp.FlatModel.SEGMENTS)
    assert max(n for _, n in ex.launches) * 4 <= 1 << 20 and sum(n for _, n in ex.launches) == flat.numel
    assert flat.ranges["visual_frontend."][1] - flat.ranges["visual_frontend."][0] < 200000      # layer1 + stem trail
    ok = True
    for i, (n, p) in enumerate(m.named_parameters()):
        want = (1.0 + (i % 7)) * (1 + 2) / (1.0 if overlap == "sum" else 2.0)       # sum / mean over ranks of (rank+1)*pattern
        ok = ok and bool(torch.allclose(p.grad, torch.full_like(p.grad, want)))
    chk = flat.flat_param.double().sum().item()
    q.put((rank, ok, chk, flat.numel))
    dist.destroy_process_group()


def _run(overlap, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-6         # replicas identical after the broadcast
    assert res[0][3] == res[1][3] >= 23000000        # 1+1-layer model: frontend 11.2 M + enc 3.4 M + dec 8.4 M


def test_dp_flat_allreduce_world2():
    _run(False, 29611)


def test_dp_overlap_hooks_world2():
    _run(True, 29612)


def test_dp_sum_mode_world2():
    _run("sum", 29613)


def test_gradient_exchange_refuses_second_backward_before_finish():
    """a segment already exchanged (averaged in place) must not silently swallow a second backward."""
    from sbl_for_multilingual_lip_reading_amd import dp

    class _F:      # the two members launch() touches
        flat_grad = torch.zeros(8)
        def segment_grad(self, seg):
            return self.flat_grad[:0]
    ex = dp.GradientExchange.__new__(dp.GradientExchange)
    ex.flat, ex.world, ex._pending, ex.cuda = _F(), 2, [], False
    ex.launch("decoder.")
    with pytest.raises(RuntimeError, match="finish"):
        ex.launch("decoder.")
    ex.finish()
    ex.launch("decoder.")
