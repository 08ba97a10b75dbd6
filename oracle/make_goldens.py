"""Golden-vector generator: runs the REFERENCE ITSELF (imported read-only from
/root/reference, never copied) on deterministic weights/inputs and writes small
.npz fixtures to tests/golden/.  Run only in the build container:

    python oracle/make_goldens.py            # all fixtures (SBL dir)
    python oracle/make_goldens.py --cls      # BASELINE config-1 fixture (CLS dir; separate
                                             # process: both dirs use the package name 'transformer')

Inputs and weights are NOT stored: they are functions of (name, shape, salt) via
sbl_for_multilingual_lip_reading_amd/detfill.py, which tests re-evaluate.
Randomness in the reference is neutralised without editing it:
every nn.Dropout.p = 0, torch.nn.functional.dropout patched to identity (for the
always-on call at SBL/transformer/video_frontend.py:122), random.seed(k) before
forward fixes the 16 teacher-forcing coins (SBL/transformer/decoder.py:176).
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbl_for_multilingual_lip_reading_amd import detfill  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF_SBL = "/root/reference/SBL_Multilingual_Lip_reading"
REF_CLS = "/root/reference/VSR_visual_frontend_pretraining_on_LRW_LRW1000_classify"


def neutralise_dropout(model):
    import torch.nn as nn
    import torch.nn.functional as F
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x


def det_load(model, salt=0, gains=None):
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith(".pe"):
            new[k] = v
        else:
            new[k] = torch.from_numpy(detfill.fill_value(k, tuple(v.shape), salt, gains).copy())
    model.load_state_dict(new)


def build_sbl(n_enc, n_dec, gains=None):
    from transformer.decoder import Decoder
    from transformer.encoder import Encoder
    from transformer.transformer import Transformer
    enc = Encoder(512, n_enc, 8, 64, 64, 512, 2048)
    dec = Decoder(0, 1, 58, 512, n_dec, 8, 64, 64, 512, 2048)
    m = Transformer(enc, dec, None)
    neutralise_dropout(m)
    det_load(m, gains=gains)
    return m


def assert_varied(tag, ids_by_dir, margins):
    """The fixture must exercise the token feedback loop (decoder.py:166-186): >= 6 distinct arg-max ids per direction, no
    two samples with the same id sequence, every arg-max margin > 1e-2 (so a 1e-3-accurate evaluation picks the same ids)."""
    for d, ids in ids_by_dir.items():
        ids = np.asarray(ids)
        assert len(set(ids.flatten().tolist())) >= 6, (tag, d, ids)
        rows = [tuple(r) for r in ids.tolist()]
        assert len(set(rows)) == len(rows), (tag, d, "two samples decode alike")
    assert float(np.min(margins)) > 1e-2, (tag, float(np.min(margins)))


FULL_GRADS = [
    "visual_frontend.frontend3D.0.weight", "visual_frontend.frontend3D.1.weight",
    "visual_frontend.frontend3D.1.bias",
    "visual_frontend.resnet18.layer1.0.bn1.weight", "visual_frontend.resnet18.layer2.0.downsample.0.weight",
    "visual_frontend.resnet18.layer4.1.bn2.bias",
    "encoder.layer_norm_in.weight", "encoder.linear_in.bias",
    "encoder.layer_stack.0.slf_attn.w_qs.bias", "encoder.layer_stack.0.pos_ffn.layer_norm.weight",
    "decoder.tgt_word_emb.weight", "decoder.tgt_word_prj_l2r.weight", "decoder.tgt_word_prj_r2l.weight",
    "decoder.layer_first_l2r.slf_attn.w_ks.bias", "decoder.layer_first_r2l.enc_attn.w_vs.bias",
    "decoder.layer_first_l2r.pos_ffn.layer_norm.bias",
]


def e2e(tag, B, T, H, W, n_enc, n_dec, coin_seed, salt, gains=None):
    """Train-mode forward + loss + backward of Transformer (SBL/train.py:188-196)."""
    from transformer.loss import cal_performance
    m = build_sbl(n_enc, n_dec, gains)
    m.train()
    x, l2r, r2l = detfill.synthetic_batch(B, T, H, W, salt)
    random.seed(coin_seed)
    coins = [random.random() > 0.5 for _ in range(16)]
    random.seed(coin_seed)
    # hooks for intermediate features
    grabbed = {}
    m.visual_frontend.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("feats", o.detach().clone()))
    m.encoder.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("enc", o[0].detach().clone()))
    pl, gl, pr, gr = m(torch.from_numpy(x), torch.from_numpy(l2r), torch.from_numpy(r2l))
    ll, ncl = cal_performance(pl, gl, smoothing=0.1)
    lr, ncr = cal_performance(pr, gr, smoothing=0.1)
    loss = 0.5 * (ll + lr)
    loss.backward()
    out = {"B": B, "T": T, "H": H, "W": W, "n_enc": n_enc, "n_dec": n_dec, "salt": salt,
           "coin_seed": coin_seed, "coins": np.array(coins),
           "feats": grabbed["feats"].numpy(), "enc": grabbed["enc"].numpy(),
           "pred_l2r": pl.detach().numpy(), "pred_r2l": pr.detach().numpy(),
           "gold_l2r": gl.numpy(), "gold_r2l": gr.numpy(),
           "loss_l2r": ll.item(), "loss_r2l": lr.item(), "loss": loss.item(),
           "n_correct_l2r": ncl, "n_correct_r2l": ncr}
    # argmax margins (top1 - top2) of every logit row: parity tests need them > tolerance
    for d, p in (("l2r", pl), ("r2l", pr)):
        top = p.detach().topk(2, dim=-1).values
        out["margin_" + d] = (top[..., 0] - top[..., 1]).numpy()
        out["argmax_" + d] = p.detach().argmax(-1).numpy()
    names, norms = [], []
    for k, p in m.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        if k in FULL_GRADS:
            out["grad:" + k] = p.grad.numpy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    # running stats after one train step (BN side effects)
    sd = m.state_dict()
    for k in ("visual_frontend.frontend3D.1.running_mean", "visual_frontend.frontend3D.1.running_var",
              "visual_frontend.resnet18.layer4.1.bn2.running_mean", "visual_frontend.resnet18.layer4.1.bn2.running_var",
              "visual_frontend.resnet18.layer2.0.downsample.1.running_var"):
        out["after:" + k] = sd[k].numpy()
    out["after:nbt"] = sd["visual_frontend.frontend3D.1.num_batches_tracked"].numpy()
    if gains is not None:
        out["gains"] = np.array(gains)
        assert_varied(tag, {"l2r": out["argmax_l2r"], "r2l": out["argmax_r2l"]}, [out["margin_l2r"].min(), out["margin_r2l"].min()])
    np.savez_compressed(os.path.join(OUT, "e2e_%s.npz" % tag), **out)
    print("e2e", tag, "loss", loss.item(), "min margin", float(min(out["margin_l2r"].min(), out["margin_r2l"].min())))


def eval_recognize(tag, B, T, H, W, n_enc, n_dec, salt, gains=None, train_bn=False):
    """Greedy decode, Transformer.recognize (transformer.py:45-69, SBL/test.py:174).  Default: eval mode (running-stat
    BN).  train_bn=True runs the same reference method with the module in train() mode (batch-statistics BatchNorm, every
    dropout p = 0): with the deterministic running statistics the encoder output is nearly the same for every sample, in
    train mode the samples differ, and so do their greedy token sequences."""
    m = build_sbl(n_enc, n_dec, gains)
    m.train() if train_bn else m.eval()
    x, l2r, r2l = detfill.synthetic_batch(B, T, H, W, salt)
    grabbed = {"logits": []}
    m.visual_frontend.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("feats", o.detach().clone()))
    m.encoder.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("enc", o[0].detach().clone()))
    for head in (m.decoder.tgt_word_prj_l2r, m.decoder.tgt_word_prj_r2l):      # arg-max margins of every greedy step
        head.register_forward_hook(lambda mod, i, o: grabbed["logits"].append(o.detach().clone()))
    with torch.no_grad():
        ys_l, ys_r = m.recognize(torch.from_numpy(x))
    margins = []
    for lg in grabbed["logits"]:
        top = lg.reshape(-1, lg.size(-1)).topk(2, -1).values
        margins.append(float((top[:, 0] - top[:, 1]).min()))
    extra = {}
    if gains is not None:
        extra["gains"] = np.array(gains)
        assert_varied(tag, {"l2r": ys_l.numpy()[:, 1:], "r2l": ys_r.numpy()[:, 1:]}, margins)
    np.savez_compressed(os.path.join(OUT, "recognize_%s.npz" % tag), B=B, T=T, H=H, W=W, n_enc=n_enc,
                        n_dec=n_dec, salt=salt, feats=grabbed["feats"].numpy(), enc=grabbed["enc"].numpy(),
                        ys_l2r=ys_l.numpy(), ys_r2l=ys_r.numpy(), min_margin=min(margins), train_bn=int(train_bn), **extra)
    print("recognize", tag, ys_l[0].tolist(), "min margin", min(margins))


def state_dict_keys():
    """The reference model's own state-dict surface (537 keys, shapes, dtypes): the drop-in contract of SURVEY 3.5 / 8b."""
    m = build_sbl(6, 6)
    sd = m.state_dict()
    keys = sorted(sd.keys())
    np.savez_compressed(os.path.join(OUT, "state_dict_keys.npz"), keys=np.array(keys),
                        shapes=np.array([",".join(str(d) for d in sd[k].shape) for k in keys]),
                        dtypes=np.array([str(sd[k].dtype) for k in keys]),
                        param_names=np.array([n for n, _ in m.named_parameters()]),
                        buffer_names=np.array([n for n, _ in m.named_buffers()]))
    print("state_dict_keys", len(keys))


def modules():
    """Module-level fixtures with their own tiny deterministic weights."""
    import transformer.attention as A
    import transformer.decoder as D
    import transformer.module as M
    import transformer.utils as U
    import transformer.video_frontend as V
    from transformer.loss import cal_performance
    out = {}
    t = lambda name, shape, s=1.0: torch.from_numpy(detfill.uniform(name, shape) * np.float32(s))

    # ScaledDotProductAttention, with / without causal mask (attention.py:63-83)
    sd_ = A.ScaledDotProductAttention(8.0, attn_dropout=0.0)
    q, k, v = t("sdpa.q", (16, 7, 64)), t("sdpa.k", (16, 11, 64)), t("sdpa.v", (16, 11, 64))
    o, a = sd_(q, k, v)
    out["sdpa.out"], out["sdpa.attn"] = o.numpy(), a.numpy()
    qs = t("sdpa.qs", (16, 9, 64))
    ks, vs = t("sdpa.ks", (16, 9, 64)), t("sdpa.vs", (16, 9, 64))
    cm = U.get_subsequent_mask(torch.zeros(16, 9, dtype=torch.long))
    o, a = sd_(qs, ks, vs, mask=cm)
    out["sdpa.causal_out"], out["sdpa.causal_attn"] = o.numpy(), a.numpy()

    # MultiHeadAttention self + cross, fwd and input/param grads (attention.py:6-60)
    mh = A.MultiHeadAttention(8, 512, 64, 64, dropout=0.0)
    det_load(mh)
    x = t("mha.x", (3, 5, 512)).requires_grad_(True)
    mem = t("mha.mem", (3, 29, 512)).requires_grad_(True)
    cm = U.get_subsequent_mask(torch.zeros(3, 5, dtype=torch.long))
    o1, a1 = mh(x, x, x, mask=cm)
    o2, a2 = mh(o1, mem, mem, mask=None)
    (o2 * t("mha.dy", (3, 5, 512))).sum().backward()
    out["mha.self_out"], out["mha.self_attn"] = o1.detach().numpy(), a1.detach().numpy()
    out["mha.cross_out"], out["mha.cross_attn"] = o2.detach().numpy(), a2.detach().numpy()
    out["mha.dx"], out["mha.dmem"] = x.grad.numpy(), mem.grad.numpy()
    for k_, p in mh.named_parameters():      # big matrices: strided subsample keeps the fixture small
        out["mha.grad:" + k_] = p.grad.numpy()[::4, ::4] if p.dim() == 2 else p.grad.numpy()

    # PositionwiseFeedForward (module.py:35-52)
    ff = M.PositionwiseFeedForward(512, 2048, dropout=0.0)
    det_load(ff)
    x = t("ffn.x", (3, 5, 512)).requires_grad_(True)
    o = ff(x)
    (o * t("ffn.dy", (3, 5, 512))).sum().backward()
    out["ffn.out"], out["ffn.dx"] = o.detach().numpy(), x.grad.numpy()
    for k_, p in ff.named_parameters():
        out["ffn.grad:" + k_] = p.grad.numpy()[::8, ::8] if p.dim() == 2 else p.grad.numpy()

    # PositionalEncoding table (module.py:8-32)
    out["pe"] = M.PositionalEncoding(512, max_len=64).pe[0].numpy()

    # DecoderLayer first (causal) and later (no self mask) (decoder.py:387-408)
    dl = D.DecoderLayer(512, 2048, 8, 64, 64, dropout=0.0)
    det_load(dl)
    x = t("dl.x", (2, 6, 512))
    mem = t("dl.mem", (2, 29, 512))
    ones = torch.ones(2, 6, 1)
    cm = U.get_subsequent_mask(torch.zeros(2, 6, dtype=torch.long))
    with torch.no_grad():
        out["dl.causal_out"] = dl(x.clone(), mem, non_pad_mask=ones, slf_attn_mask=cm)[0].numpy()
        out["dl.plain_out"] = dl(x.clone(), mem, non_pad_mask=ones, slf_attn_mask=None)[0].numpy()

    # SBL fusion: the reference's aliased in-place loops, verbatim semantics (decoder.py:127-143)
    a = t("fus.a", (2, 6, 512)).clone()
    b = t("fus.b", (2, 6, 512)).clone()
    L = a.size(1)
    l2r_index = list(range(L)); r2l_index = list(range(L - 1, -1, -1))
    left, right = a, b
    for n in range(L):
        left[:, n] = a[:, l2r_index[n]] + b[:, r2l_index[n]]
    for n in range(L):
        right[:, n] = b[:, l2r_index[n]] + a[:, r2l_index[n]]
    out["fus.a_out"], out["fus.b_out"] = left.numpy(), right.numpy()

    # preprocess / pad_list (decoder.py:62-77)
    dec = D.Decoder(0, 1, 58, 512, 1, 8, 64, 64, 512, 2048)
    _, l2r, r2l = detfill.synthetic_batch(4, 2, 2, 2, 3)
    yi, yo = dec.preprocess(torch.from_numpy(l2r))
    out["prep.tgt"], out["prep.ys_in"], out["prep.ys_out"] = l2r, yi.numpy(), yo.numpy()

    # loss, smoothing 0.1 and 0, with some IGNORE_ID golds (loss.py:7-52)
    pred = t("loss.pred", (4, 16, 58), 3.0).requires_grad_(True)
    gold = torch.from_numpy(((detfill.uniform("loss.gold", (4, 16)) + 1) * 29).astype(np.int64).clip(0, 57))
    gold[1, 10:] = -1
    gold[3, 5:] = -1
    for sm, nm in ((0.1, "ls"), (0.0, "ce")):
        pred.grad = None
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            l, nc = cal_performance(pred, gold, smoothing=sm)
        l.backward()
        out["loss.%s" % nm], out["loss.%s_ncorrect" % nm] = l.item(), nc
        out["loss.%s_dpred" % nm] = pred.grad.numpy().copy()
    out["loss.gold"] = gold.numpy()

    # stem + trunk on a tiny clip, train mode, with input-free grads (video_frontend.py:91-125)
    fe = V.Lipreading()
    neutralise_dropout(fe)
    sdv = fe.state_dict()
    fe.load_state_dict({k_: torch.from_numpy(detfill.fill_value("visual_frontend." + k_, tuple(v_.shape)).copy())
                        for k_, v_ in sdv.items()})
    fe.train()
    x = torch.from_numpy(detfill.normal("fe.x", (2, 1, 6, 32, 32)))
    grabbed = {}
    fe.frontend3D.register_forward_hook(lambda mod, i, o: grabbed.__setitem__("stem", o.detach().clone()))
    fe.frontend3D[0].register_forward_hook(lambda mod, i, o: grabbed.__setitem__("conv", o.detach().clone()))
    y = fe(x)
    (y * t("fe.dy", (2, 6, 512))).sum().backward()
    out["fe.conv"], out["fe.stem"], out["fe.out"] = grabbed["conv"].numpy(), grabbed["stem"].numpy(), y.detach().numpy()
    for k_ in ("frontend3D.0.weight", "frontend3D.1.weight", "frontend3D.1.bias", "resnet18.layer1.0.conv1.weight",
               "resnet18.layer1.0.bn1.weight", "resnet18.layer2.0.conv1.weight", "resnet18.layer2.0.downsample.0.weight",
               "resnet18.layer3.0.downsample.1.bias"):
        out["fe.grad:" + k_] = dict(fe.named_parameters())[k_].grad.numpy()
    out["fe.after:frontend3D.1.running_mean"] = fe.state_dict()["frontend3D.1.running_mean"].numpy()
    out["fe.after:frontend3D.1.running_var"] = fe.state_dict()["frontend3D.1.running_var"].numpy()
    fe.eval()
    with torch.no_grad():
        out["fe.eval_out"] = fe(x).numpy()

    # Noam schedule + Adam (optimizer.py:1-27; SBL/train.py:75)
    from transformer.optimizer import TransformerOptimizer
    p = torch.nn.Parameter(t("opt.p", (257,)))
    opt = TransformerOptimizer(torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.98), eps=1e-09))
    lrs = []
    for s in range(3):
        opt.zero_grad()
        p.grad = t("opt.g%d" % s, (257,), 0.01)
        opt.step()
        lrs.append(opt.lr)
    out["opt.p_after3"], out["opt.lrs"] = p.detach().numpy(), np.array(lrs)
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **out)
    print("modules: %d arrays" % len(out))


def cls():
    """BASELINE config 1: the reference's CLS frontend + Encoder + heads, composed
    as its evident intent (the shipped forward raises)."""
    sys.path.insert(0, REF_CLS)
    from transformer.encoder import Encoder
    from transformer.transformer import Transformer
    m = Transformer(Encoder(512, 6, 8, 64, 64, 512, 2048), None)
    neutralise_dropout(m)
    det_load(m)
    m.train()
    x, _, _ = detfill.synthetic_batch(2, 29, 88, 88, 11)
    xt = torch.from_numpy(x)
    xt = torch.cat([xt, xt.new_zeros(2, 1, 88, 88)], 1)
    feats = m.visual_frontend(xt.view(2, 1, 30, 88, 88))
    enc, *_ = m.encoder_v(feats, [30, 30])
    v = m.fc_1500(enc.mean(1))
    lang = m.fc_2(enc[:, -1])
    np.savez_compressed(os.path.join(OUT, "cls_config1.npz"), salt=11, feats=feats.detach().numpy(),
                        enc=enc.detach().numpy(), v_t=v.detach().numpy(), v_lang=lang.detach().numpy())
    print("cls", v.shape, lang.shape)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cls", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if args.cls:
        cls()
        sys.exit(0)
    sys.path.insert(0, REF_SBL)
    if args.only in ("", "modules"):
        modules()
    if args.only in ("", "small"):
        e2e("small", 2, 8, 24, 24, 2, 2, coin_seed=3, salt=5)
    if args.only in ("", "full"):
        e2e("full", 2, 29, 88, 88, 6, 6, coin_seed=7, salt=7)
    if args.only in ("", "rec"):
        eval_recognize("full", 2, 29, 88, 88, 6, 6, salt=7)
        eval_recognize("small", 2, 8, 24, 24, 2, 2, salt=5)
    # fixtures whose arg-max ids vary over steps, samples and directions (they exercise the token feedback loop)
    if args.only in ("", "varied"):
        e2e("varied", 3, 29, 88, 88, 6, 6, coin_seed=13, salt=34, gains="varied")
        eval_recognize("varied", 3, 29, 88, 88, 6, 6, salt=34, gains="varied", train_bn=True)
    if args.only in ("", "keys"):
        state_dict_keys()
