"""CPU oracle for the SBL lip-reading forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under sbl_for_multilingual_lip_reading_amd/
may import this file; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker / the CPU number printed beside
the GPU one.  The product path is the HIP library behind include/sbl_hip.h.

What it is: a from-scratch *functional* restatement, in plain fp32 PyTorch CPU
ops, of the algorithm of /root/reference/SBL_Multilingual_Lip_reading/transformer
(abbreviated SBL/ below).  It takes a flat {state-dict name: tensor} dict with
the reference's 537 key names, so reference weights drop in unchanged.  It is
written differently from the reference on purpose (closed-form SBL fusion,
cross-attention K/V hoisted out of the 16-step loop, explicit coin list,
explicit dropout switch) — those are exactly the algebraic rewrites the HIP path
uses, so the oracle also proves them equal to the reference.

Pinned by: tests/golden/*.npz, produced by oracle/make_goldens.py from the
*reference itself* (imported from /root/reference in the build container) with
deterministic weights/inputs; tests/test_oracle_golden.py checks this file
against every one of them (forward values, loss, greedy tokens, gradients).
The reference ships no tests or golden vectors of its own.

Gradients come from torch.autograd over these CPU ops.
"""
import math
import random

import torch
import torch.nn.functional as F

IGNORE_ID = -1          # SBL/config.py:25
MAXLEN = 16             # SBL/transformer/decoder.py:95, utils.py:5
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# visual frontend: SBL/transformer/video_frontend.py:10-125
# --------------------------------------------------------------------------- #
def _bn(sd, prefix, x, training):
    """nn.BatchNorm{2,3}d with torch defaults (video_frontend.py:21,24,71,101).
    In training mode running stats in `sd` are updated in place and
    num_batches_tracked is incremented, like the module does."""
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def stem(sd, x, training, prefix="visual_frontend.frontend3D"):
    """Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3)) -> BN3d -> ReLU -> MaxPool3d
    ((1,3,3),s(1,2,2),p(0,1,1)); video_frontend.py:99-104.
    x: (N,1,T,H,W) -> (N,64,T,H/4,W/4)."""
    y = F.conv3d(x, sd[prefix + ".0.weight"], None, (1, 2, 2), (2, 3, 3))
    y = F.relu(_bn(sd, prefix + ".1", y, training))
    return F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))


def basic_block(sd, prefix, x, stride, training):
    """video_frontend.py:15-41 (conv3x3-BN-ReLU-conv3x3-BN-(+shortcut)-ReLU)."""
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(sd, prefix + ".bn1", out, training))
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, prefix + ".bn2", out, training)
    if (prefix + ".downsample.0.weight") in sd:
        res = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride, 0)
        res = _bn(sd, prefix + ".downsample.1", res, training)
    else:
        res = x
    return F.relu(out + res)


def trunk(sd, x, training, prefix="visual_frontend.resnet18"):
    """ResNet-18 without stem, [2,2,2,2] blocks, global avg-pool;
    video_frontend.py:44-89.  x: (N*T,64,h,w) -> (N*T,512)."""
    for li in range(1, 5):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = basic_block(sd, "%s.layer%d.%d" % (prefix, li, bi), x, stride, training)
    return x.mean(dim=(2, 3))


def frontend(sd, x, training=True, drop_mask=None):
    """Lipreading.forward, video_frontend.py:111-125.  x: (N,1,T,H,W).
    drop_mask: None = the always-on F.dropout(p=0.5) (video_frontend.py:122) is
    neutralised (parity mode); else a {0,1} float tensor (N*T,512) applied as
    x * mask * 2."""
    N, _, T = x.shape[:3]
    y = stem(sd, x, training)
    y = y.transpose(1, 2).contiguous().view(-1, 64, y.size(3), y.size(4))
    y = trunk(sd, y, training)
    if drop_mask is not None:
        y = y * drop_mask * 2.0
    return y.view(N, T, 512)


# --------------------------------------------------------------------------- #
# transformer primitives: SBL/transformer/attention.py, module.py
# --------------------------------------------------------------------------- #
def positional_encoding(length, d_model=512):
    """module.py:14-24: sin/cos table built in log space, fp32."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _dropout(x, p, on):
    return F.dropout(x, p, True) if on else x


def sdpa(q, k, v, mask=None, temperature=8.0):
    """ScaledDotProductAttention.forward, attention.py:72-83 (dropout off).
    q: (G,Lq,d) k,v: (G,Lk,d); mask: bool (G,Lq,Lk), True = masked."""
    attn = torch.bmm(q, k.transpose(1, 2)) / temperature
    if mask is not None:
        attn = attn.masked_fill(mask, float("-inf"))
    attn = torch.softmax(attn, dim=2)
    return torch.bmm(attn, v), attn


def _split_heads(x, n_head):
    B, L, D = x.shape
    d = D // n_head
    return x.view(B, L, n_head, d).permute(2, 0, 1, 3).reshape(n_head * B, L, d)


def _merge_heads(x, n_head):
    G, L, d = x.shape
    B = G // n_head
    return x.view(n_head, B, L, d).permute(1, 2, 0, 3).reshape(B, L, n_head * d)


def mha(sd, prefix, q_in, kv_in, mask=None, n_head=8, kv_proj=None, drop=0.0):
    """MultiHeadAttention.forward, attention.py:32-60.
    mask: bool (B,Lq,Lk) or None.  kv_proj: optional pre-projected (K,V) of
    kv_in (the hoisted cross-attention projections).  Returns (out, attn)."""
    q = F.linear(q_in, sd[prefix + ".w_qs.weight"], sd[prefix + ".w_qs.bias"])
    if kv_proj is None:
        kv_proj = mha_project_kv(sd, prefix, kv_in)
    k, v = kv_proj
    qh, kh, vh = _split_heads(q, n_head), _split_heads(k, n_head), _split_heads(v, n_head)
    m = None if mask is None else mask.repeat(n_head, 1, 1)
    attn = torch.bmm(qh, kh.transpose(1, 2)) / math.sqrt(qh.size(-1))
    if m is not None:
        attn = attn.masked_fill(m, float("-inf"))
    attn = torch.softmax(attn, dim=2)
    o = _merge_heads(torch.bmm(_dropout(attn, drop, drop > 0), vh), n_head)
    o = _dropout(F.linear(o, sd[prefix + ".fc.weight"], sd[prefix + ".fc.bias"]), drop, drop > 0)
    o = F.layer_norm(o + q_in, (q_in.size(-1),), sd[prefix + ".layer_norm.weight"],
                     sd[prefix + ".layer_norm.bias"], LN_EPS)
    return o, attn


def mha_project_kv(sd, prefix, kv_in):
    k = F.linear(kv_in, sd[prefix + ".w_ks.weight"], sd[prefix + ".w_ks.bias"])
    v = F.linear(kv_in, sd[prefix + ".w_vs.weight"], sd[prefix + ".w_vs.bias"])
    return k, v


def ffn(sd, prefix, x, drop=0.0):
    """PositionwiseFeedForward.forward, module.py:47-52."""
    h = F.relu(F.linear(x, sd[prefix + ".w_1.weight"], sd[prefix + ".w_1.bias"]))
    o = _dropout(F.linear(h, sd[prefix + ".w_2.weight"], sd[prefix + ".w_2.bias"]), drop, drop > 0)
    return F.layer_norm(o + x, (x.size(-1),), sd[prefix + ".layer_norm.weight"],
                        sd[prefix + ".layer_norm.bias"], LN_EPS)


# --------------------------------------------------------------------------- #
# encoder: SBL/transformer/encoder.py
# --------------------------------------------------------------------------- #
def encoder(sd, x, n_layers=6, n_head=8, prefix="encoder", drop=0.0, return_attns=False):
    """Encoder.forward, encoder.py:36-67, for full-length inputs
    (transformer.py:37 => masks are all-false / all-ones, encoder.py:86,89 no-ops).
    x: (N,T,d_input) -> (N,T,512)."""
    T = x.size(1)
    h = F.linear(x, sd[prefix + ".linear_in.weight"], sd[prefix + ".linear_in.bias"])
    h = F.layer_norm(h, (h.size(-1),), sd[prefix + ".layer_norm_in.weight"],
                     sd[prefix + ".layer_norm_in.bias"], LN_EPS)
    h = _dropout(h + positional_encoding(T, h.size(-1)).unsqueeze(0), drop, drop > 0)
    attns = []
    for n in range(n_layers):
        p = "%s.layer_stack.%d" % (prefix, n)
        h, a = mha(sd, p + ".slf_attn", h, h, None, n_head, drop=drop)
        h = ffn(sd, p + ".pos_ffn", h, drop)
        attns.append(a)
    return (h, attns) if return_attns else h


# --------------------------------------------------------------------------- #
# SBL decoder: SBL/transformer/decoder.py
# --------------------------------------------------------------------------- #
def preprocess(padded, sos_id=0, eos_id=1):
    """Decoder.preprocess, decoder.py:62-77: strip -1, prepend sos / append eos,
    pad both to 16 with *eos* (utils.py:1-9).  Returns (ys_in, ys_out) (N,16)."""
    N = padded.size(0)
    ys_in = padded.new_full((N, MAXLEN), eos_id)
    ys_out = padded.new_full((N, MAXLEN), eos_id)
    for b in range(N):
        y = padded[b][padded[b] != IGNORE_ID]
        n = y.numel()
        ys_in[b, 0] = sos_id
        ys_in[b, 1:1 + n] = y
        ys_out[b, :n] = y
    return ys_in, ys_out


def sbl_fusion(a, b):
    """Closed form of the aliased in-place slice loops decoder.py:132-143 and
    :160-164:  A' = A + flip(B),
    B' = B + flip(A') = 2B + flip(A);  flip along the prefix (time) axis."""
    a2 = a + b.flip(1)
    b2 = b + a2.flip(1)
    return a2, b2


def decoder_layer(sd, prefix, x, enc, slf_mask, kv_proj, n_head=8, drop=0.0):
    """DecoderLayer.forward, decoder.py:396-408 (non_pad_mask is all ones)."""
    x, _ = mha(sd, prefix + ".slf_attn", x, x, slf_mask, n_head, drop=drop)
    x, _ = mha(sd, prefix + ".enc_attn", x, enc, None, n_head, kv_proj=kv_proj, drop=drop)
    return ffn(sd, prefix + ".pos_ffn", x, drop)


def _decoder_steps(sd, enc, n_layers, n_head, next_token, drop=0.0, prefix="decoder"):
    """The 16-step loop shared by Decoder.forward (decoder.py:106-186) and
    recognize_beam (decoder.py:310-383).  next_token(i, pred_l2r, pred_r2l) ->
    (tok_l2r, tok_r2l) picks the fed-back tokens."""
    N = enc.size(0)
    names = {"l2r": ["%s.layer_first_l2r" % prefix] + ["%s.layer_stack_l2r.%d" % (prefix, n) for n in range(n_layers - 1)],
             "r2l": ["%s.layer_first_r2l" % prefix] + ["%s.layer_stack_r2l.%d" % (prefix, n) for n in range(n_layers - 1)]}
    # cross-attention K/V of the encoder output are step-invariant: hoist (exact algebra)
    kv = {d: [mha_project_kv(sd, p + ".enc_attn", enc) for p in names[d]] for d in names}
    emb = sd[prefix + ".tgt_word_emb.weight"]
    pe = positional_encoding(MAXLEN + 1, emb.size(1))
    ys_l = torch.zeros(N, 1, dtype=torch.long)      # sos_id = 0
    ys_r = torch.zeros(N, 1, dtype=torch.long)
    out_l, out_r = [], []
    for i in range(MAXLEN):
        L = i + 1
        causal = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1).unsqueeze(0).expand(N, -1, -1)
        a = _dropout(F.embedding(ys_l, emb) + pe[:L].unsqueeze(0), drop, drop > 0)
        b = _dropout(F.embedding(ys_r, emb) + pe[:L].unsqueeze(0), drop, drop > 0)
        for n in range(n_layers):
            m = causal if n == 0 else None          # decoder.py:123-125 vs :150,:157
            a = decoder_layer(sd, names["l2r"][n], a, enc, m, kv["l2r"][n], n_head, drop)
            b = decoder_layer(sd, names["r2l"][n], b, enc, m, kv["r2l"][n], n_head, drop)
            a, b = sbl_fusion(a, b)
        pred_l = F.linear(a[:, -1], sd[prefix + ".tgt_word_prj_l2r.weight"])
        pred_r = F.linear(b[:, -1], sd[prefix + ".tgt_word_prj_r2l.weight"])
        out_l.append(pred_l)
        out_r.append(pred_r)
        tl, tr = next_token(i, pred_l, pred_r)
        ys_l = torch.cat([ys_l, tl.view(N, 1)], 1)
        ys_r = torch.cat([ys_r, tr.view(N, 1)], 1)
    return torch.stack(out_l, 1), torch.stack(out_r, 1), ys_l, ys_r


def draw_coins(n=MAXLEN):
    """The per-step teacher-forcing coins of decoder.py:176, in the reference's
    own order of python `random` draws: True => feed own argmax."""
    return [random.random() > 0.5 for _ in range(n)]


def decoder_forward(sd, tgt_l2r, tgt_r2l, enc, coins, n_layers=6, n_head=8, drop=0.0):
    """Decoder.forward, decoder.py:79-191.  coins: list of 16 bools
    (True = own argmax, False = gold[:, i]).  Returns
    (pred_l2r (N,16,58), gold_l2r (N,16), pred_r2l, gold_r2l, ys_l2r (N,17), ys_r2l)."""
    _, gold_l = preprocess(tgt_l2r)
    _, gold_r = preprocess(tgt_r2l)

    def nxt(i, pl, pr):
        if coins[i]:
            return pl.argmax(-1), pr.argmax(-1)
        return gold_l[:, i], gold_r[:, i]

    pl, pr, ys_l, ys_r = _decoder_steps(sd, enc, n_layers, n_head, nxt, drop)
    return pl, gold_l, pr, gold_r, ys_l, ys_r


def recognize_beam(sd, enc, n_layers=6, n_head=8):
    """Decoder.recognize_beam, decoder.py:301-385: greedy, always own argmax.
    Returns (ys_l2r, ys_r2l) (N,17) int64."""
    _, _, ys_l, ys_r = _decoder_steps(sd, enc, n_layers, n_head,
                                      lambda i, pl, pr: (pl.argmax(-1), pr.argmax(-1)))
    return ys_l, ys_r


# --------------------------------------------------------------------------- #
# loss: SBL/transformer/loss.py
# --------------------------------------------------------------------------- #
def cal_performance(pred, gold, smoothing=0.0):
    """loss.py:7-52.  Label-smoothed CE with q = onehot*(1-eps) + (1-onehot)*eps/C
    (rows do NOT sum to 1 — kept), mean over gold != -1; n_correct over the same."""
    pred = pred.reshape(-1, pred.size(-1))
    gold = gold.reshape(-1)
    valid = gold.ne(IGNORE_ID)
    if smoothing > 0.0:
        C = pred.size(1)
        one_hot = torch.zeros_like(pred).scatter(1, (gold * valid.long()).view(-1, 1), 1)
        q = one_hot * (1 - smoothing) + (1 - one_hot) * smoothing / C
        loss = -(q * F.log_softmax(pred, dim=1)).sum(1)
        loss = loss.masked_select(valid).sum() / valid.sum()
    else:
        loss = F.cross_entropy(pred, gold, ignore_index=IGNORE_ID, reduction="mean")
    n_correct = pred.argmax(1).eq(gold).masked_select(valid).sum().item()
    return loss, n_correct


# --------------------------------------------------------------------------- #
# whole model: SBL/transformer/transformer.py, SBL/train.py:188-196
# --------------------------------------------------------------------------- #
def transformer_forward(sd, x, tgt_l2r, tgt_r2l, coins, n_layers_enc=6, n_layers_dec=6,
                        n_head=8, training=True, drop=0.0, frontend_mask=None):
    """Transformer.forward, transformer.py:22-43.  x: (N,T,H,W)."""
    feats = frontend(sd, x.unsqueeze(1), training, frontend_mask)
    enc = encoder(sd, feats, n_layers_enc, n_head, drop=drop)
    pl, gl, pr, gr, ys_l, ys_r = decoder_forward(sd, tgt_l2r, tgt_r2l, enc, coins,
                                                 n_layers_dec, n_head, drop)
    return {"feats": feats, "enc": enc, "pred_l2r": pl, "gold_l2r": gl,
            "pred_r2l": pr, "gold_r2l": gr, "ys_l2r": ys_l, "ys_r2l": ys_r}


def train_step_loss(out, smoothing=0.1):
    """SBL/train.py:190-193: loss = 0.5*(loss_l2r + loss_r2l)."""
    ll, _ = cal_performance(out["pred_l2r"], out["gold_l2r"], smoothing)
    lr, _ = cal_performance(out["pred_r2l"], out["gold_r2l"], smoothing)
    return 0.5 * (ll + lr)


def recognize(sd, x, n_layers_enc=6, n_layers_dec=6, n_head=8, training=False):
    """Transformer.recognize, transformer.py:45-69 (frontend dropout neutralised)."""
    feats = frontend(sd, x.unsqueeze(1), training, None)
    enc = encoder(sd, feats, n_layers_enc, n_head)
    return recognize_beam(sd, enc, n_layers_dec, n_head)


def cls_forward(sd, x, n_layers_enc=6, n_head=8, training=True):
    """BASELINE config 1 (CLS pre-training plumbing).  The shipped
    CLS/transformer/transformer.py:16-37 cannot run; this
    is the documented restatement of its evident intent: x (N,T,H,W) is padded by
    one zero frame, frontend -> encoder ("encoder_v.*" keys) ->
    fc_1500(mean over time), fc_2(last frame)."""
    x = torch.cat([x, x.new_zeros(x.size(0), 1, x.size(2), x.size(3))], 1)
    feats = frontend(sd, x.unsqueeze(1), training, None)
    enc = encoder(sd, feats, n_layers_enc, n_head, prefix="encoder_v")
    v = F.linear(enc.mean(1), sd["fc_1500.weight"], sd["fc_1500.bias"])
    lang = F.linear(enc[:, -1], sd["fc_2.weight"], sd["fc_2.bias"])
    return feats, enc, v, lang


# --------------------------------------------------------------------------- #
# optimizer: SBL/transformer/optimizer.py, SBL/train.py:75
# --------------------------------------------------------------------------- #
def noam_lr(step_num, k=0.2, warmup_steps=4000, d_model=512):
    """optimizer.py:22-27."""
    return k * d_model ** (-0.5) * min(step_num ** (-0.5), step_num * warmup_steps ** (-1.5))


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-9):
    """torch.optim.Adam (no weight decay, no amsgrad) as SBL/train.py:75
    configures it; in place on fp32 tensors."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------- #
# input pipeline: SBL/data_gen.py:122-125,104-108,276-296; SBL/cvtransforms.py:7-48  (numpy, like the reference)
# The reference's cvtransforms.py imports cv2 (absent here), so these few lines are restated, not imported:
# "parity unpinned" by a reference run; the arithmetic is ColorNormalize's one formula in float64.
# --------------------------------------------------------------------------- #
def preprocess_clip_ref(frames_u8, y1, x1, flip, removed, Tout=30, crop=(88, 88)):
    """frames_u8: (Tin,Hin,Win) uint8.  removed: iterable of frame indices i>0 that FrameRemoval replaces by frame
    i-1 (sequentially, data_gen.py:104-108).  Returns float32 (Tout, th, tw)."""
    import numpy as np
    vid = frames_u8.astype(np.float64) / 255.                      # load_file
    vid = (vid - 0.413621) / 0.1700239                             # ColorNormalize
    th, tw = crop
    vid = vid[:, y1:y1 + th, x1:x1 + tw].copy()                    # RandomCrop / CenterCrop
    if flip:
        vid = vid[:, :, ::-1].copy()                               # HorizontalFlip (cv2.flip(img, 1))
    for i in range(vid.shape[0]):                                  # FrameRemoval
        if i in removed and i > 0:
            vid[i] = vid[i - 1]
    out = np.zeros((Tout, th, tw), dtype=np.float32)               # vids[:length] = vid
    out[:vid.shape[0]] = vid
    return out


# --------------------------------------------------------------------------- #
# helpers used by tests / bench
# --------------------------------------------------------------------------- #
def state_dict_shapes(n_layers_enc=6, n_layers_dec=6, d_input=512, d_model=512, d_inner=2048,
                      n_head=8, d_k=64, d_v=64, vocab=58):
    """{name: shape} for every learned/buffered entry of the reference SBL model
    except the two computed 'pe' buffers."""
    s = {}

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    s["visual_frontend.frontend3D.0.weight"] = (64, 1, 5, 7, 7)
    bn("visual_frontend.frontend3D.1", 64)
    inp = 64
    for li, planes in enumerate([64, 128, 256, 512], 1):
        for bi in range(2):
            p = "visual_frontend.resnet18.layer%d.%d" % (li, bi)
            s[p + ".conv1.weight"] = (planes, inp, 3, 3); bn(p + ".bn1", planes)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn2", planes)
            if bi == 0 and li > 1:
                s[p + ".downsample.0.weight"] = (planes, inp, 1, 1); bn(p + ".downsample.1", planes)
            inp = planes

    def lin(p, o, i, bias=True):
        s[p + ".weight"] = (o, i)
        if bias:
            s[p + ".bias"] = (o,)

    def ln(p, d):
        s[p + ".weight"] = (d,); s[p + ".bias"] = (d,)

    def mha_(p):
        lin(p + ".w_qs", n_head * d_k, d_model); lin(p + ".w_ks", n_head * d_k, d_model)
        lin(p + ".w_vs", n_head * d_v, d_model); ln(p + ".layer_norm", d_model)
        lin(p + ".fc", d_model, n_head * d_v)

    def ffn_(p):
        lin(p + ".w_1", d_inner, d_model); lin(p + ".w_2", d_model, d_inner); ln(p + ".layer_norm", d_model)

    lin("encoder.linear_in", d_model, d_input); ln("encoder.layer_norm_in", d_model)
    for n in range(n_layers_enc):
        mha_("encoder.layer_stack.%d.slf_attn" % n); ffn_("encoder.layer_stack.%d.pos_ffn" % n)
    s["decoder.tgt_word_emb.weight"] = (vocab, d_model)

    def dl(p):
        mha_(p + ".slf_attn"); mha_(p + ".enc_attn"); ffn_(p + ".pos_ffn")

    dl("decoder.layer_first_l2r")
    for n in range(n_layers_dec - 1):
        dl("decoder.layer_stack_l2r.%d" % n)
    dl("decoder.layer_first_r2l")
    for n in range(n_layers_dec - 1):
        dl("decoder.layer_stack_r2l.%d" % n)
    lin("decoder.tgt_word_prj_l2r", 58, 512, bias=False)
    lin("decoder.tgt_word_prj_r2l", 58, 512, bias=False)
    return s


def make_state_dict(n_layers_enc=6, n_layers_dec=6, salt=0, requires_grad=False, gains=None):
    """Deterministically filled oracle state dict (same routine the GPU tests use
    for the HIP-backed modules)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sbl_for_multilingual_lip_reading_amd import detfill
    vals = detfill.fill_state_dict(state_dict_shapes(n_layers_enc, n_layers_dec), salt, gains)
    sd = {}
    for k, v in vals.items():
        t = torch.from_numpy(v.copy())
        if requires_grad and t.is_floating_point() and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd
