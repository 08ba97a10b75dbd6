"""The SBL decoder's 16 steps as ONE autograd node with a stage-batched backward.

Forward has to run stage by stage: a stage (a maximal run of teacher-forced steps, see Decoder._run) needs the
previous stage's argmax token (decoder.py:173-186).  Backward has no such dependency - tokens are not differentiable -
so the backward of ALL 16 steps is one ragged batch: per layer and direction one LayerNorm / attention / GEMM launch
over the N*136 rows of every step, instead of one per stage.  The dependent chain shrinks from
11 kernels x 6 layers x (1 + #own-argmax coins) to 11 x 6, and its GEMMs are 4352-row products instead of 100-1500.

To make that possible every stage's forward writes its activations into row ranges of per-layer buffers laid out
as the 16-segment ragged batch (segment t = the step with prefix length t+1, rows N*t*(t+1)/2 ...), with the raw
C-ABI kernels (no per-op tape).  Dropout masks are functions of (seed, stream offset, element index); a stage passes
an offset that folds its row base in (the generator is affine in both, see _fold), so the batched backward
regenerates exactly the masks of the forward with plain whole-buffer indices.

Used by Decoder.forward when gradients accumulate into persistent buffers (dp.FlatModel) and the coins are known on
the host; every other case keeps the per-stage tape (Decoder._run).  Same numbers as that path (tests compare them).
"""
import torch

from ._env import config, ops

_C1 = 0x9E3779B97F4A7C15
_C2 = 0xD1B54A32D192ED03
_K = (_C2 * pow(_C1, -1, 1 << 64)) % (1 << 64)
_MASK = (1 << 64) - 1


def _fold(offset, base):
    """Stream offset that makes element i of a sub-tensor draw what element base+i of the whole tensor draws:
    sbl_rand_u32 hashes seed + C1*(offset+1) + C2*idx in wrap-around 64-bit arithmetic and C1 is odd."""
    return ((offset + 1 + base * _K) & _MASK) - 1 & _MASK


def _p(t):
    return None if t is None else t.data_ptr()


def supported(dec, encoder_outputs):
    """The fast path needs persistent gradient buffers for every decoder parameter (kernels accumulate there), host
    coins, CUDA, and the 6-layer / 512-wide geometry the buffers are laid out for."""
    if not (dec.batched_backward and encoder_outputs.is_cuda and torch.is_grad_enabled() and dec.coins_dev is None):
        return False
    if dec.d_model != 512 or not dec.batch_teacher_runs:
        return False
    return all(ops._gbuf(p) is not None for p in dec.parameters())


class _Layer:
    """Parameter handles of one DecoderLayer (fused QKV weights of the self-attention are adjacent rows)."""

    def __init__(self, lay):
        lay.slf_attn._fuse()
        lay.enc_attn._fuse()
        sa, ea, ff = lay.slf_attn, lay.enc_attn, lay.pos_ffn
        self.wqkv, self.bqkv = sa.w_qs.weight, sa.w_qs.bias                  # first of three adjacent blocks
        assert ops._adjacent(sa.w_qs.weight, sa.w_ks.weight, sa.w_vs.weight) and ops._adjacent(sa.w_qs.bias, sa.w_ks.bias, sa.w_vs.bias)
        gw = [ops._gbuf(t) for t in (sa.w_qs.weight, sa.w_ks.weight, sa.w_vs.weight)]
        gb = [ops._gbuf(t) for t in (sa.w_qs.bias, sa.w_ks.bias, sa.w_vs.bias)]
        assert ops._adjacent(*gw) and ops._adjacent(*gb)
        self.g_wqkv, self.g_bqkv = gw[0], gb[0]
        self.wfc_s, self.bfc_s, self.g_wfc_s, self.g_bfc_s = sa.fc.weight, sa.fc.bias, ops._gbuf(sa.fc.weight), ops._gbuf(sa.fc.bias)
        self.ln_s = (sa.layer_norm.weight, sa.layer_norm.bias, ops._gbuf(sa.layer_norm.weight), ops._gbuf(sa.layer_norm.bias), sa.layer_norm.eps)
        self.wq, self.bq, self.g_wq, self.g_bq = ea.w_qs.weight, ea.w_qs.bias, ops._gbuf(ea.w_qs.weight), ops._gbuf(ea.w_qs.bias)
        assert ops._adjacent(ea.w_ks.weight, ea.w_vs.weight) and ops._adjacent(ea.w_ks.bias, ea.w_vs.bias)
        self.wkv, self.bkv = ea.w_ks.weight, ea.w_ks.bias
        self.wv_e, self.bv_e = ea.w_vs.weight, ea.w_vs.bias
        gkw, gkb = [ops._gbuf(t) for t in (ea.w_ks.weight, ea.w_vs.weight)], [ops._gbuf(t) for t in (ea.w_ks.bias, ea.w_vs.bias)]
        assert ops._adjacent(*gkw) and ops._adjacent(*gkb)
        self.g_wkv, self.g_bkv = gkw[0], gkb[0]
        self.g_wv_e, self.g_bv_e = gkw[1], gkb[1]
        self.wfc_e, self.bfc_e, self.g_wfc_e, self.g_bfc_e = ea.fc.weight, ea.fc.bias, ops._gbuf(ea.fc.weight), ops._gbuf(ea.fc.bias)
        self.ln_e = (ea.layer_norm.weight, ea.layer_norm.bias, ops._gbuf(ea.layer_norm.weight), ops._gbuf(ea.layer_norm.bias), ea.layer_norm.eps)
        self.w1, self.b1, self.g_w1, self.g_b1 = ff.w_1.weight, ff.w_1.bias, ops._gbuf(ff.w_1.weight), ops._gbuf(ff.w_1.bias)
        self.w2, self.b2, self.g_w2, self.g_b2 = ff.w_2.weight, ff.w_2.bias, ops._gbuf(ff.w_2.weight), ops._gbuf(ff.w_2.bias)
        self.ln_f = (ff.layer_norm.weight, ff.layer_norm.bias, ops._gbuf(ff.layer_norm.weight), ops._gbuf(ff.layer_norm.bias), ff.layer_norm.eps)
        self.drop_s = sa.dropout.p if sa.training else 0.0
        self.drop_e = ea.dropout.p if ea.training else 0.0
        self.drop_f = ff.dropout.p if ff.training else 0.0


class DecoderStagesFn(torch.autograd.Function):
    """(encoder_outputs, anchor parameter) -> (pred_l2r, pred_r2l), each (N, 16, 58).  `anchor` is any decoder
    parameter: it only makes the outputs require grad when the encoder is frozen."""

    @staticmethod
    def forward(ctx, enc_out, anchor, dec, gold_l2r, gold_r2l, coins):
        call, gemm, segs = ops.call, ops.gemm, ops._segs
        dev = enc_out.device
        N, T, D = enc_out.shape
        H, HD, F_, V = 8, 512, dec.layer_first_l2r.pos_ffn.w_1.weight.size(0), dec.tgt_word_emb.weight.size(0)
        ML = config.MAX_DECODE_LEN
        nl = dec.n_layers
        R = N * ML * (ML + 1) // 2
        rowoff = [N * t * (t + 1) // 2 for t in range(ML + 1)]                        # segment t = prefix length t+1
        ps_off = [H * N * sum((u + 1) ** 2 for u in range(t)) for t in range(ML + 1)]  # self-attention probabilities
        pe_off = [H * N * T * sum(u + 1 for u in range(t)) for t in range(ML + 1)]     # cross-attention probabilities
        layers = [[_Layer(l) for l in dec._layers(d)] for d in (0, 1)]
        enc2 = enc_out.contiguous().view(N * T, D)
        main = torch.cuda.current_stream(dev)
        side = ops.side_stream(dev) if dec.two_streams else None
        ops.set_main_stream(main)
        streams = (main, side if side is not None else main)
        st = ops.dropout_state(dev)
        training = dec.training
        p_emb = dec.dropout.p if training else 0.0

        def E(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)

        # ---- all-stage buffers
        B_ = [[None] * nl for _ in (0, 1)]
        for d in (0, 1):
            for n in range(nl):
                B_[d][n] = dict(x=E(R, D), qkv=E(R, 3 * HD), att=E(R, HD), ps=E(ps_off[ML]), o_s=E(R, D), mu_s=E(R), rs_s=E(R), y_s=E(R, D),
                                q=E(R, HD), att2=E(R, HD), pe=E(pe_off[ML]), o_e=E(R, D), mu_e=E(R), rs_e=E(R), y_e=E(R, D),
                                h=E(R, F_), o_f=E(R, D), mu_f=E(R), rs_f=E(R), y_f=E(R, D), kv=None,
                                off=[st.next_offset() for _ in range(5)] if training else [0] * 5)
        lazy_io = not (getattr(dec, "fuse_stage_io", True) and D == 512 and V <= 64)
        xout = [E(R, D), E(R, D)] if lazy_io else [None, None]      # (the fused stage tail never materialises the last fusion)
        x0 = [E(R, D) if (p_emb > 0 and lazy_io) else None for _ in (0, 1)]        # pre-dropout embeddings are not needed afterwards
        off_emb = [st.next_offset() if p_emb > 0 else 0 for _ in (0, 1)]
        last = [E(ML * N, D), E(ML * N, D)]
        pred = [E(ML * N, V), E(ML * N, V)]
        heads = (dec.tgt_word_prj_l2r.weight, dec.tgt_word_prj_r2l.weight)
        emb, pe_tab = dec.tgt_word_emb.weight, dec.positional_encoding.pe[0]
        golds = (gold_l2r, gold_r2l)
        # token buffers: <sos> followed by the teacher tokens of every step (decoder.py:176-186 feeds gold[:, i] when coin i says
        # so); the slots of the own-arg-max steps are overwritten by the stage tail before any later step reads them.  One
        # concatenation per direction instead of a fill, a column store and one select launch per teacher-forced step.
        sos_col = torch.full((N, 1), dec.sos_id, dtype=torch.long, device=dev)
        ys = [torch.cat([sos_col, golds_[:, :ML].to(torch.long)], 1).contiguous() for golds_ in (gold_l2r, gold_r2l)]
        seed = st.seed

        # ---- hoisted cross-attention K/V (attention.py:42-43 for every layer and direction; step-invariant).  With the flat
        # parameter layout of dp.FlatModel the 12 [W_k; W_v] pairs are rows of ONE (12*1024, 512) matrix: one GEMM writes
        # KV_all (N*T, 12*1024), layer (d, n) reads its column block in place (row stride ldkv).  Otherwise one launch per
        # layer for both directions, or one GEMM per layer and direction with r2l on the side stream.
        merged = getattr(dec, "merge_directions", True) and all(
            (layers[0][n].drop_s, layers[0][n].drop_e, layers[0][n].drop_f, layers[0][n].ln_s[4], layers[0][n].ln_e[4], layers[0][n].ln_f[4]) ==
            (layers[1][n].drop_s, layers[1][n].drop_e, layers[1][n].drop_f, layers[1][n].ln_s[4], layers[1][n].ln_e[4], layers[1][n].ln_f[4])
            for n in range(nl))
        kv_order = [layers[d][n] for d in (0, 1) for n in range(nl)]
        kv_fused = (getattr(dec, "fuse_kv_projections", True)
                    and ops._adjacent(*[w for L in kv_order for w in (L.wkv, L.wv_e)])
                    and ops._adjacent(*[b for L in kv_order for b in (L.bkv, L.bv_e)])
                    and ops._adjacent(*[g for L in kv_order for g in (L.g_wkv, L.g_wv_e)])
                    and ops._adjacent(*[g for L in kv_order for g in (L.g_bkv, L.g_bv_e)]))
        ldkv = 2 * HD * (2 * nl if kv_fused else 1)
        if kv_fused:
            kv_all = E(N * T, ldkv)
            gemm(0, 1, N * T, ldkv, D, enc2, D, kv_order[0].wkv, D, kv_all, ldkv, bias=kv_order[0].bkv)
            for i, (d, n) in enumerate((d, n) for d in (0, 1) for n in range(nl)):
                B_[d][n]["kv"] = kv_all[:, i * 2 * HD:(i + 1) * 2 * HD]
        elif merged:
            for n in range(nl):
                L0, L1 = layers[0][n], layers[1][n]
                B_[0][n]["kv"], B_[1][n]["kv"] = E(N * T, 2 * HD), E(N * T, 2 * HD)
                ops.gemm2(N * T, 2 * HD, D, enc2, enc2, D, L0.wkv, L1.wkv, D, B_[0][n]["kv"], B_[1][n]["kv"], 2 * HD, L0.bkv, L1.bkv)
        else:
            if side is not None:
                side.wait_stream(main)
            for d in (0, 1):
                with torch.cuda.stream(streams[d]):
                    for n in range(nl):
                        L = layers[d][n]
                        B_[d][n]["kv"] = E(N * T, 2 * HD)
                        gemm(0, 1, N * T, 2 * HD, D, enc2, D, L.wkv, D, B_[d][n]["kv"], 2 * HD, bias=L.bkv)

        # ---- stages (same rule as Decoder._run)
        stages, i = [], 0
        while i < ML:
            j = i
            while j < ML - 1 and not coins[j]:
                j += 1
            stages.append((i, j))
            i = j + 1

        def layer_fwd(d, n, r0, r1, i0, segL):
            L, b = layers[d][n], B_[d][n]
            M = r1 - r0
            seg_arr, nseg = segs(segL)
            x = b["x"][r0:r1]
            # self-attention sub-layer
            qkv = b["qkv"][r0:r1]
            gemm(0, 1, M, 3 * HD, D, x, D, L.wqkv, D, qkv, 3 * HD, bias=L.bqkv)
            call("sbl_attention_seg_fwd", _p(qkv), 3 * HD, _p(qkv[:, HD:]), 3 * HD, _p(qkv[:, 2 * HD:]), 3 * HD, _p(b["att"][r0:r1]), HD,
                 b["ps"].data_ptr() + 4 * ps_off[i0], 1 if n == 0 else 0, None, N, H, seg_arr, nseg, 0, 0.125, L.drop_s,
                 _p(seed) if L.drop_s > 0 else None, _fold(b["off"][0], ps_off[i0]), ops._s())
            gemm(0, 1, M, D, HD, b["att"][r0:r1], HD, L.wfc_s, HD, b["o_s"][r0:r1], D, bias=L.bfc_s)
            g, be, _, _, eps = L.ln_s
            call("sbl_add_layernorm_fwd", _p(b["o_s"][r0:r1]), _p(x), _p(g), _p(be), _p(b["y_s"][r0:r1]), _p(b["mu_s"][r0:r1]),
                 _p(b["rs_s"][r0:r1]), M, D, eps, L.drop_s, _p(seed) if L.drop_s > 0 else None, _fold(b["off"][1], r0 * D), ops._s())
            # cross-attention sub-layer
            y_s, q = b["y_s"][r0:r1], b["q"][r0:r1]
            gemm(0, 1, M, HD, D, y_s, D, L.wq, D, q, HD, bias=L.bq)
            kv = b["kv"]
            call("sbl_attention_seg_fwd", _p(q), HD, _p(kv), ldkv, _p(kv[:, HD:]), ldkv, _p(b["att2"][r0:r1]), HD,
                 b["pe"].data_ptr() + 4 * pe_off[i0], 0, None, N, H, seg_arr, nseg, T, 0.125, L.drop_e,
                 _p(seed) if L.drop_e > 0 else None, _fold(b["off"][2], pe_off[i0]), ops._s())
            gemm(0, 1, M, D, HD, b["att2"][r0:r1], HD, L.wfc_e, HD, b["o_e"][r0:r1], D, bias=L.bfc_e)
            g, be, _, _, eps = L.ln_e
            call("sbl_add_layernorm_fwd", _p(b["o_e"][r0:r1]), _p(y_s), _p(g), _p(be), _p(b["y_e"][r0:r1]), _p(b["mu_e"][r0:r1]),
                 _p(b["rs_e"][r0:r1]), M, D, eps, L.drop_e, _p(seed) if L.drop_e > 0 else None, _fold(b["off"][3], r0 * D), ops._s())
            # position-wise feed-forward sub-layer
            y_e, h = b["y_e"][r0:r1], b["h"][r0:r1]
            gemm(0, 1, M, F_, D, y_e, D, L.w1, D, h, F_, bias=L.b1, relu=1)
            gemm(0, 1, M, D, F_, h, F_, L.w2, F_, b["o_f"][r0:r1], D, bias=L.b2)
            g, be, _, _, eps = L.ln_f
            call("sbl_add_layernorm_fwd", _p(b["o_f"][r0:r1]), _p(y_e), _p(g), _p(be), _p(b["y_f"][r0:r1]), _p(b["mu_f"][r0:r1]),
                 _p(b["rs_f"][r0:r1]), M, D, eps, L.drop_f, _p(seed) if L.drop_f > 0 else None, _fold(b["off"][4], r0 * D), ops._s())

        def layer_fwd2(n, r0, r1, i0, segL, fuse_next=False):
            """Both directions of layer n in shared launches (same shapes, their own operands): kernel boundaries cost
            ~5 us each and small launches on two streams do not overlap, so the directions share launches, not streams."""
            L0, L1, b0, b1 = layers[0][n], layers[1][n], B_[0][n], B_[1][n]
            M = r1 - r0
            seg_arr, nseg = segs(segL)
            sl = lambda b, k: b[k][r0:r1]
            sp = _p(seed) if training else None
            # self-attention sub-layer
            q0, q1 = sl(b0, "qkv"), sl(b1, "qkv")
            ops.gemm2(M, 3 * HD, D, sl(b0, "x"), sl(b1, "x"), D, L0.wqkv, L1.wqkv, D, q0, q1, 3 * HD, L0.bqkv, L1.bqkv)
            call("sbl_attention_seg2_fwd", _p(q0), _p(q1), 3 * HD, _p(q0[:, HD:]), _p(q1[:, HD:]), 3 * HD, _p(q0[:, 2 * HD:]), _p(q1[:, 2 * HD:]),
                 3 * HD, _p(sl(b0, "att")), _p(sl(b1, "att")), HD, b0["ps"].data_ptr() + 4 * ps_off[i0], b1["ps"].data_ptr() + 4 * ps_off[i0],
                 1 if n == 0 else 0, N, H, seg_arr, nseg, 0, 0.125, L0.drop_s, sp if L0.drop_s > 0 else None,
                 _fold(b0["off"][0], ps_off[i0]), _fold(b1["off"][0], ps_off[i0]), ops._s())
            ops.gemm2(M, D, HD, sl(b0, "att"), sl(b1, "att"), HD, L0.wfc_s, L1.wfc_s, HD, sl(b0, "o_s"), sl(b1, "o_s"), D, L0.bfc_s, L1.bfc_s)

            def ln2(o, res, y, mu, rs, ln0, ln1, drop_p, k):
                call("sbl_add_layernorm2_fwd", _p(sl(b0, o)), _p(sl(b1, o)), _p(sl(b0, res)), _p(sl(b1, res)), _p(ln0[0]), _p(ln1[0]),
                     _p(ln0[1]), _p(ln1[1]), _p(sl(b0, y)), _p(sl(b1, y)), _p(sl(b0, mu)), _p(sl(b1, mu)), _p(sl(b0, rs)), _p(sl(b1, rs)),
                     M, D, ln0[4], drop_p, sp if drop_p > 0 else None, _fold(b0["off"][k], r0 * D), _fold(b1["off"][k], r0 * D), ops._s())

            ln2("o_s", "x", "y_s", "mu_s", "rs_s", L0.ln_s, L1.ln_s, L0.drop_s, 1)
            # cross-attention sub-layer
            ops.gemm2(M, HD, D, sl(b0, "y_s"), sl(b1, "y_s"), D, L0.wq, L1.wq, D, sl(b0, "q"), sl(b1, "q"), HD, L0.bq, L1.bq)
            kv0, kv1 = b0["kv"], b1["kv"]
            call("sbl_attention_seg2_fwd", _p(sl(b0, "q")), _p(sl(b1, "q")), HD, _p(kv0), _p(kv1), ldkv, _p(kv0[:, HD:]), _p(kv1[:, HD:]), ldkv,
                 _p(sl(b0, "att2")), _p(sl(b1, "att2")), HD, b0["pe"].data_ptr() + 4 * pe_off[i0], b1["pe"].data_ptr() + 4 * pe_off[i0],
                 0, N, H, seg_arr, nseg, T, 0.125, L0.drop_e, sp if L0.drop_e > 0 else None,
                 _fold(b0["off"][2], pe_off[i0]), _fold(b1["off"][2], pe_off[i0]), ops._s())
            ops.gemm2(M, D, HD, sl(b0, "att2"), sl(b1, "att2"), HD, L0.wfc_e, L1.wfc_e, HD, sl(b0, "o_e"), sl(b1, "o_e"), D, L0.bfc_e, L1.bfc_e)
            ln2("o_e", "y_s", "y_e", "mu_e", "rs_e", L0.ln_e, L1.ln_e, L0.drop_e, 3)
            # position-wise feed-forward sub-layer
            ops.gemm2(M, F_, D, sl(b0, "y_e"), sl(b1, "y_e"), D, L0.w1, L1.w1, D, sl(b0, "h"), sl(b1, "h"), F_, L0.b1, L1.b1, relu=1)
            ops.gemm2(M, D, F_, sl(b0, "h"), sl(b1, "h"), F_, L0.w2, L1.w2, F_, sl(b0, "o_f"), sl(b1, "o_f"), D, L0.b2, L1.b2)
            if fuse_next:
                # the sub-layer's LayerNorm and the cross-direction fusion that feeds layer n + 1, one launch; y_f is not stored
                nb0, nb1 = B_[0][n + 1]["x"][r0:r1], B_[1][n + 1]["x"][r0:r1]
                call("sbl_add_layernorm2_fusion_fwd", _p(sl(b0, "o_f")), _p(sl(b1, "o_f")), _p(sl(b0, "y_e")), _p(sl(b1, "y_e")),
                     _p(L0.ln_f[0]), _p(L1.ln_f[0]), _p(L0.ln_f[1]), _p(L1.ln_f[1]), _p(nb0), _p(nb1), _p(sl(b0, "mu_f")), _p(sl(b1, "mu_f")),
                     _p(sl(b0, "rs_f")), _p(sl(b1, "rs_f")), N, seg_arr, nseg, D, L0.ln_f[4], L0.drop_f, sp if L0.drop_f > 0 else None,
                     _fold(b0["off"][4], r0 * D), _fold(b1["off"][4], r0 * D), ops._s())
            else:
                ln2("o_f", "y_e", "y_f", "mu_f", "rs_f", L0.ln_f, L1.ln_f, L0.drop_f, 4)

        fused_io = getattr(dec, "fuse_stage_io", True) and D == 512 and V <= 64
        for (i0, i1) in stages:
            segL = tuple(range(i0 + 1, i1 + 2))
            seg_arr, nseg = segs(segL)
            r0, r1 = rowoff[i0], rowoff[i1 + 1]
            M = r1 - r0
            if fused_io:
                # stage head: embedding + PE + dropout of both directions in one launch (the pre-dropout embeddings are not
                # needed afterwards: backward regenerates the mask)
                call("sbl_embed_pe_drop2_fwd", _p(ys[0]), _p(ys[1]), ys[0].stride(0), _p(emb), _p(pe_tab), _p(B_[0][0]["x"][r0:r1]),
                     _p(B_[1][0]["x"][r0:r1]), N, seg_arr, nseg, D, V, p_emb, _p(seed) if p_emb > 0 else None,
                     _fold(off_emb[0], r0 * D), _fold(off_emb[1], r0 * D), ops._s())
            else:
                for d in (0, 1):
                    dst = x0[d][r0:r1] if p_emb > 0 else B_[d][0]["x"][r0:r1]
                    call("sbl_embed_pe_seg_fwd", _p(ys[d]), ys[d].stride(0), _p(emb), _p(pe_tab), _p(dst), N, seg_arr, nseg, D, V, ops._s())
                    if p_emb > 0:
                        call("sbl_dropout", _p(dst), _p(B_[d][0]["x"][r0:r1]), M * D, p_emb, _p(seed), _fold(off_emb[d], r0 * D), ops._s())
            for n in range(nl):
                fuse_next = merged and fused_io and n + 1 < nl
                if merged:
                    layer_fwd2(n, r0, r1, i0, segL, fuse_next)
                else:
                    if side is not None:
                        side.wait_stream(main)
                    for d in (0, 1):
                        with torch.cuda.stream(streams[d]):
                            layer_fwd(d, n, r0, r1, i0, segL)
                    if side is not None:
                        main.wait_stream(side)
                if (n + 1 == nl and fused_io) or fuse_next:
                    continue       # the last fusion is only ever read at the last positions (stage tail below); the others
                                   # rode on the LayerNorm launch
                nxt = [B_[d][n + 1]["x"][r0:r1] if n + 1 < nl else xout[d][r0:r1] for d in (0, 1)]
                call("sbl_fusion_seg_fwd", _p(B_[0][n]["y_f"][r0:r1]), _p(B_[1][n]["y_f"][r0:r1]), _p(nxt[0]), _p(nxt[1]), N, seg_arr, nseg,
                     D, ops._s())
            lrows = slice(i0 * N, (i1 + 1) * N)
            if fused_io:
                # stage tail: last fusion at the last positions + both heads + the token fed to the next stage, one launch
                call("sbl_decoder_tail_fwd", _p(B_[0][nl - 1]["y_f"][r0:r1]), _p(B_[1][nl - 1]["y_f"][r0:r1]), _p(heads[0]), _p(heads[1]),
                     _p(last[0][lrows]), _p(last[1][lrows]), _p(pred[0][lrows]), _p(pred[1][lrows]), V, _p(ys[0]), _p(ys[1]),
                     ys[0].stride(0), i1, int(bool(coins[i1])), N, seg_arr, nseg, D, V, ops._s())
                continue
            for d in (0, 1):
                call("sbl_gather_last_fwd", _p(xout[d][r0:r1]), _p(last[d][lrows]), N, seg_arr, nseg, D, ops._s())
                gemm(0, 1, nseg * N, V, D, last[d][lrows], D, heads[d], D, pred[d][lrows], V)
            if coins[i1]:
                for d in (0, 1):
                    ops.argmax_select(pred[d][i1 * N:(i1 + 1) * N], golds[d], ys[d], i1, 1)

        ctx.state = dict(N=N, T=T, D=D, H=H, HD=HD, F=F_, V=V, ML=ML, nl=nl, R=R, layers=layers, B=B_, xout=xout, last=last, ys=ys,
                         heads=heads, g_heads=(ops._gbuf(heads[0]), ops._gbuf(heads[1])), g_emb=ops._gbuf(emb), seed=seed,
                         p_emb=p_emb, off_emb=off_emb, streams=streams, two=side is not None, enc2=enc2, training=training,
                         kv_fused=kv_fused, ldkv=ldkv, kv_order=kv_order, defer_wgrads=bool(getattr(dec, "defer_weight_grads", False)))
        ctx.set_materialize_grads(False)
        dec.last_ys = ys
        # (ML*N, V) step-major -> (N, ML, V) views
        return pred[0].view(ML, N, V).transpose(0, 1), pred[1].view(ML, N, V).transpose(0, 1)

    @staticmethod
    @ops._bw
    def backward(ctx, dpl, dpr):
        S = ctx.state
        call, gemm, segs = ops.call, ops.gemm, ops._segs
        N, T, D, H, HD, F_, V, ML, nl, R = (S[k] for k in ("N", "T", "D", "H", "HD", "F", "V", "ML", "nl", "R"))
        layers, B_, streams = S["layers"], S["B"], S["streams"]
        main, side = streams[0], (streams[1] if S["two"] else None)
        dev = S["enc2"].device
        seed = S["seed"]
        segL = tuple(range(1, ML + 1))
        seg_arr, nseg = segs(segL)
        kv_fused, ldkv, kv_order = S["kv_fused"], S["ldkv"], S["kv_order"]
        # gradient of the hoisted K/V: column block i = (direction, layer) of one (N*T, 12*1024) buffer when the projections
        # are fused (its input gradient is then ONE product over K = 12*1024 below), else one buffer per layer and direction
        dkv_all_buf = None

        def E(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)

        wg = []                                        # deferred weight gradients: (C, ldc, colsum, A, lda, B, ldb, M, N)

        def dW(C, colsum, A, lda, Bm, ldb, M, Nn):
            wg.append((C, Nn, colsum, A, lda, Bm, ldb, M, Nn))

        if side is not None:
            side.wait_stream(main)
        # ---- heads and the gather of the last positions
        dx = [None, None]
        for d, dp in ((0, dpl), (1, dpr)):
            with torch.cuda.stream(streams[d]):
                if dp is None:
                    dx[d] = torch.zeros(R, D, device=dev, dtype=torch.float32)
                    continue
                dpred = dp.transpose(0, 1).contiguous().view(ML * N, V)
                dlast = E(ML * N, D)
                gemm(0, 0, ML * N, D, V, dpred, V, S["heads"][d], D, dlast, D)
                gemm(1, 0, V, D, ML * N, dpred, V, S["last"][d], D, S["g_heads"][d], D, accumulate=1)
                dx[d] = E(R, D)
                call("sbl_gather_last_bwd", _p(dlast), _p(dx[d]), N, seg_arr, nseg, D, ops._s())

        def ln_bwd(dy, o, res, ln, mu, rs, drop_p, off):
            g, _, gg, gb, _ = ln
            dz = E(R, D)
            do = E(R, D)          # separate: the deferred weight gradient reads it after dz has been accumulated into
            call("sbl_add_layernorm_bwd", _p(dy), _p(o), _p(res), _p(g), _p(mu), _p(rs), _p(dz), _p(do), _p(gg), _p(gb), R, D, drop_p,
                 _p(seed) if drop_p > 0 else None, off, ops._s())
            return dz, do

        keep = []                                      # operands of the deferred weight gradients stay alive until the flush
        dkv_all = [[None] * nl for _ in (0, 1)]

        def layer_bwd(d, n, dy):
            L, b = layers[d][n], B_[d][n]
            # feed-forward
            dz, do = ln_bwd(dy, b["o_f"], b["y_e"], L.ln_f, b["mu_f"], b["rs_f"], L.drop_f, b["off"][4])
            dW(L.g_w2, L.g_b2, do, D, b["h"], F_, D, F_)
            dh = E(R, F_)
            gemm(0, 0, R, F_, D, do, D, L.w2, F_, dh, F_, mask=b["h"], ldm=F_)
            dW(L.g_w1, L.g_b1, dh, F_, b["y_e"], D, F_, D)
            gemm(0, 0, R, D, F_, dh, F_, L.w1, D, dz, D, accumulate=1)
            keep.extend((do, dh))
            # cross-attention
            dz2, do2 = ln_bwd(dz, b["o_e"], b["y_s"], L.ln_e, b["mu_e"], b["rs_e"], L.drop_e, b["off"][3])
            dW(L.g_wfc_e, L.g_bfc_e, do2, D, b["att2"], HD, D, HD)
            datt = E(R, HD)
            gemm(0, 0, R, HD, D, do2, D, L.wfc_e, HD, datt, HD)
            dq = E(R, HD)
            kv = b["kv"]
            dkv = dkv_all_buf[:, (d * nl + n) * 2 * HD:(d * nl + n + 1) * 2 * HD] if kv_fused else E(N * T, 2 * HD)
            # the 16 segments share the keys: one workgroup per (batch, head) sums their dK / dV contributions in LDS
            call("sbl_attention_seg_bwd", _p(datt), HD, _p(b["q"]), HD, _p(kv), ldkv, _p(kv[:, HD:]), ldkv, _p(b["pe"]), _p(dq), HD,
                 _p(dkv), ldkv, _p(dkv[:, HD:]), ldkv, N, H, seg_arr, nseg, T, 0.125, L.drop_e,
                 _p(seed) if L.drop_e > 0 else None, b["off"][2], ops._s())
            dkv_all[d][n] = dkv
            dW(L.g_wq, L.g_bq, dq, HD, b["y_s"], D, HD, D)
            gemm(0, 0, R, D, HD, dq, HD, L.wq, D, dz2, D, accumulate=1)
            keep.extend((do2, dq))
            # self-attention
            dz3, do3 = ln_bwd(dz2, b["o_s"], b["x"], L.ln_s, b["mu_s"], b["rs_s"], L.drop_s, b["off"][1])
            dW(L.g_wfc_s, L.g_bfc_s, do3, D, b["att"], HD, D, HD)
            datt = E(R, HD)
            gemm(0, 0, R, HD, D, do3, D, L.wfc_s, HD, datt, HD)
            dqkv = E(R, 3 * HD)
            qkv = b["qkv"]
            call("sbl_attention_seg_bwd", _p(datt), HD, _p(qkv), 3 * HD, _p(qkv[:, HD:]), 3 * HD, _p(qkv[:, 2 * HD:]), 3 * HD, _p(b["ps"]),
                 _p(dqkv), 3 * HD, _p(dqkv[:, HD:]), 3 * HD, _p(dqkv[:, 2 * HD:]), 3 * HD, N, H, seg_arr, nseg, 0, 0.125, L.drop_s,
                 _p(seed) if L.drop_s > 0 else None, b["off"][0], ops._s())
            dW(L.g_wqkv, L.g_bqkv, dqkv, 3 * HD, b["x"], D, 3 * HD, D)
            gemm(0, 0, R, D, 3 * HD, dqkv, 3 * HD, L.wqkv, D, dz3, D, accumulate=1)
            keep.extend((do3, dqkv))
            return dz3

        if kv_fused:
            dkv_all_buf = E(N * T, ldkv)
        for n in range(nl - 1, -1, -1):
            if side is not None:
                main.wait_stream(side)
            dy = [E(R, D), E(R, D)]
            call("sbl_fusion_seg_bwd", _p(dx[0]), _p(dx[1]), _p(dy[0]), _p(dy[1]), N, seg_arr, nseg, D, ops._s())
            if side is not None:
                side.wait_stream(main)
            for d in (0, 1):
                with torch.cuda.stream(streams[d]):
                    dx[d] = layer_bwd(d, n, dy[d])
        # ---- embeddings (shared table: float atomics) and the hoisted K/V projections
        denc = [E(N * T, D), None if kv_fused else E(N * T, D)]
        for d in (0, 1):
            with torch.cuda.stream(streams[d]):
                g = dx[d]
                if S["p_emb"] > 0:
                    g2 = E(R, D)
                    call("sbl_dropout", _p(g), _p(g2), R * D, S["p_emb"], _p(seed), S["off_emb"][d], ops._s())
                    g = g2
                call("sbl_embed_seg_bwd", _p(S["ys"][d]), S["ys"][d].stride(0), _p(g), _p(S["g_emb"]), N, seg_arr, nseg, D, V, ops._s())
                if not kv_fused:
                    for n in range(nl):
                        L = layers[d][n]
                        dkv = dkv_all[d][n]
                        gemm(0, 0, N * T, D, 2 * HD, dkv, 2 * HD, L.wkv, D, denc[d], D, accumulate=1 if n else 0)
                        wg.append((L.g_wkv, D, L.g_bkv, dkv, 2 * HD, S["enc2"], D, 2 * HD, D))   # rows = N*T: its own group
        if side is not None:
            main.wait_stream(side)
        if kv_fused:
            # dEnc = [dKV_0 | ... | dKV_11] (N*T, 12*1024) x [Wkv_0; ...; Wkv_11] (12*1024, 512): ONE product instead of 12
            # launches plus the adds of their partial results; its weight gradient is one (12*1024, 512) problem too
            gemm(0, 0, N * T, D, ldkv, dkv_all_buf, ldkv, kv_order[0].wkv, D, denc[0], D)
            wg.append((kv_order[0].g_wkv, D, kv_order[0].g_bkv, dkv_all_buf, ldkv, S["enc2"], D, ldkv, D))
        else:
            denc[0].add_(denc[1])
        # ---- every weight gradient of the decoder: one grouped launch per row count (R rows; N*T rows for K/V)
        deferred = _DeferredWeightGrads(wg, {id(L.g_wkv): N * T for d in (0, 1) for L in layers[d]}, R, main, side)
        # Issued right here (the grouped launch then runs beside the encoder backward, a dependent chain of 928-row products
        # that cannot fill the chip), or - dec.defer_weight_grads = True - when backward has passed the encoder.  Same-box A/B
        # of the whole step: 31.46 ms here against 31.95 ms deferred: deferring makes the encoder's own launches 2-3x faster
        # (they no longer share the CUs with 2688 tiles) but leaves the chip idle under them, and the grouped launch then
        # competes with the throughput-bound frontend backward instead.  Capping the grouped launch's grid (sbl_set_tuning
        # knob 6) to 128 / 192 workgroups beside the encoder backward: 33.69 / 32.01 ms.
        if S.get("defer_wgrads", False):
            deferred.arm()
        else:
            deferred.flush()
        ctx.state = None
        return denc[0].view(N, T, D), None, None, None, None, None


class _DeferredWeightGrads:
    """The decoder's collected weight gradients (grouped launches on the side stream).  flush() issues them; arm() postpones
    that until backward has passed the ENCODER (optional, dec.defer_weight_grads; measured slower for the whole step, see
    DecoderStagesFn.backward): beside the 2688-tile grouped launch the encoder's 928-row products run at a tenth of their
    speed (6-28 TF, profiles/r02_bench_launch_shapes.txt), but without it the chip idles under them.  Triggers of an armed
    instance, whichever comes first: ops.flush_deferred() - called from the hook on the encoder INPUT's gradient (encoder.py) and by
    dp.GradientExchange before it all-reduces the decoder segment (N > 1: the exchange must see finished gradients, so
    there the flush stays where it was) - or the end-of-backward engine callback (frozen encoder)."""

    def __init__(self, wg, rows_of, R, main, side):
        self.args = (wg, rows_of, R, main, side)
        self.device_index = main.device_index

    def arm(self):
        ops._armed.setdefault(self.device_index, []).append(self)
        torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        if self.args is None:
            return
        args, self.args = self.args, None
        lst = ops._armed.get(self.device_index, [])
        if self in lst:
            lst.remove(self)
        _flush_weight_grads(*args)


def _flush_weight_grads(wg, rows_of, R, main, side):
    """C (+)= A^T B for every collected (C, ldc, colsum, A, lda, B, ldb, M, N); grouped launches (one per distinct row
    count) on the side stream when the rows are a multiple of 16, per-weight split-K GEMMs otherwise."""
    import ctypes as ct
    run = side if side is not None else main
    if run is not main:
        run.wait_stream(main)
    groups = {}
    for e in wg:
        groups.setdefault(rows_of.get(id(e[0]), R), []).append(e)
    with torch.cuda.stream(run):
        for rows, ents in groups.items():
            if rows % 16 == 0 and len(ents) > 1 and all(e[7] % 4 == 0 and e[8] % 4 == 0 for e in ents):
                n = len(ents)
                need = ops._lib.load().sbl_wgrad_group_table_bytes(n)
                key = (run.device_index, run.cuda_stream, "stages", rows)
                tab = ops._group_tables.get(key)
                if tab is None or tab.numel() < need:
                    tab = ops._group_tables[key] = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=torch.device("cuda", run.device_index))
                ops.call("sbl_wgrad_group_f32", n, 1, (ct.c_int * 1)(rows), (ct.c_void_p * n)(*[e[3].data_ptr() for e in ents]),
                         (ct.c_long * n)(*[e[4] for e in ents]), (ct.c_void_p * n)(*[e[5].data_ptr() for e in ents]),
                         (ct.c_long * n)(*[e[6] for e in ents]), (ct.c_int * n)(*[e[7] for e in ents]), (ct.c_int * n)(*[e[8] for e in ents]),
                         (ct.c_void_p * n)(*[e[0].data_ptr() for e in ents]), (ct.c_long * n)(*[e[1] for e in ents]),
                         (ct.c_void_p * n)(*[_p(e[2]) for e in ents]), tab.data_ptr(), tab.numel(), ops._s())
            else:
                for (C, ldc, colsum, A, lda, Bm, ldb, M, Nn) in ents:
                    ops.gemm(1, 0, M, Nn, rows, A, lda, Bm, ldb, C, ldc, accumulate=1, colsum=colsum)
            for e in ents:
                e[3].record_stream(run)
                e[5].record_stream(run)
    if run is not main:
        # nothing downstream reads these gradients before the step ends (dp.GradientExchange.launch waits for the side
        # stream itself): join at the end of backward instead of stalling the main stream for the 4.5 ms grouped launch
        # (a kernel trace showed the encoder backward waiting for it)
        ops._arm_side_join()
