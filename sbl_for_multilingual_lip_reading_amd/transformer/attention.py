"""Mirror of SBL_Multilingual_Lip_reading/transformer/attention.py (same class names, constructor and
forward signatures, state-dict keys); compute is libsbl_hip.so."""
import numpy as np
import torch
import torch.nn as nn

from ._env import ops


def _as_mask(mask, B, Lq, Lk):
    return ops._mask_args(mask, B, Lq, Lk)


class MultiHeadAttention(nn.Module):
    ''' Multi-Head Attention module (attention.py:6-60).

    forward(q, k, v, mask=None) -> (output, attn), attn shaped (n_head*B, Lq, Lk) like the reference.
    Extra, optional: mask='causal' (device-side triu, no mask tensor); kv_proj = pre-projected [K|V] of `k`
    (the decoder hoists it out of its 16-step loop).  The three projection weights are kept as adjacent rows of
    one fused (3*n_head*d_k, d_model) buffer so q/k/v come from a single MFMA GEMM; state-dict keys and shapes
    are unchanged (w_qs/w_ks/w_vs .weight/.bias).
    '''

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        if d_k != 64 or d_v != 64:
            raise ValueError("the HIP attention kernel is built for d_k = d_v = 64 (SBL/utils.py:94-97)")
        self.n_head = n_head
        self.d_k = d_k
        self.d_v = d_v

        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_v)))

        self.attention = ScaledDotProductAttention(temperature=np.power(d_k, 0.5), attn_dropout=dropout)
        self.layer_norm = nn.LayerNorm(d_model)

        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)

        self.dropout = nn.Dropout(dropout)

    def _fuse(self):
        """Re-point w_qs/w_ks/w_vs parameters at adjacent rows of one buffer (idempotent, cheap pointer check).
        Needed after construction, .to(device) or any external re-allocation of the parameters."""
        ws = (self.w_qs.weight, self.w_ks.weight, self.w_vs.weight)
        bs = (self.w_qs.bias, self.w_ks.bias, self.w_vs.bias)
        if ops._adjacent(*ws) and ops._adjacent(*bs):
            return
        if getattr(ws[0], "_sbl_flat", None) is not None:
            # parameters of a dp.FlatModel are never re-allocated here.  Its layout keeps self-attention triples adjacent;
            # the decoder's cross-attention modules have (w_ks, w_vs) adjacent inside the all-layers K/V block instead,
            # which is all the cross-attention path (KVProjectFn / decoder_stages) needs
            assert ops._adjacent(*ws[1:]) and ops._adjacent(*bs[1:]), "flat model: K/V projection weights are not adjacent"
            return
        with torch.no_grad():
            fw = torch.cat([w.data for w in ws], 0).contiguous()
            fb = torch.cat([b.data for b in bs], 0).contiguous()
            if fw.is_cuda:      # one-time set-up: the old storages are released below, so the copies must have run
                torch.cuda.current_stream(fw.device).synchronize()
            r = 0
            for w, b in zip(ws, bs):
                n = w.size(0)
                w.data = fw[r:r + n]
                b.data = fb[r:r + n]
                r += n

    def project_kv(self, k):
        """[K | V] = k [W_k; W_v]^T + [b_k; b_v], shape (B*Lk, 2*n_head*64) — attention.py:42-43 for both at once."""
        self._fuse()
        return ops.KVProjectFn.apply(k.reshape(-1, k.size(-1)), self.w_ks.weight, self.w_ks.bias,
                                     self.w_vs.weight, self.w_vs.bias)

    def forward_rows(self, x2, B, segL, mask=None, kv_proj=None):
        """Sub-layer on a ragged batch of rows: x2 (B*sum(segL), d_model); segment s holds B sequences of length
        segL[s] (the SBL decoder batches the steps of one teacher-forced run this way).  kv_proj=None: self-attention
        inside each segment; else cross-attention to the pre-projected [K|V] rows (B*Lk, 2*n_head*64).
        mask: None | 'causal' (| a (B,Lq,Lk) tensor when there is a single segment).  Returns (out rows, attn flat)."""
        self._fuse()
        drop_p = self.dropout.p if self.training else 0.0
        ln = self.layer_norm
        if isinstance(mask, str) or mask is None:
            mask_kind, mask_t = (1, None) if mask == "causal" else (0, None)
        else:
            assert len(segL) == 1
            Lk = segL[0] if kv_proj is None else kv_proj.size(0) // B
            mask_kind, mask_t = _as_mask(mask, B, segL[0], Lk)
        if kv_proj is None:
            return ops.MHAFn.apply(x2, None, self.w_qs.weight, self.w_qs.bias, self.w_ks.weight, self.w_ks.bias,
                                   self.w_vs.weight, self.w_vs.bias, self.fc.weight, self.fc.bias, ln.weight, ln.bias,
                                   self.n_head, mask_kind, mask_t, drop_p, ln.eps, B, tuple(segL))
        return ops.MHAFn.apply(x2, kv_proj, self.w_qs.weight, self.w_qs.bias, None, None, None, None,
                               self.fc.weight, self.fc.bias, ln.weight, ln.bias,
                               self.n_head, mask_kind, mask_t, drop_p, ln.eps, B, tuple(segL))

    def forward(self, q, k, v, mask=None, kv_proj=None):
        sz_b, len_q, d = q.size()
        len_k = k.size(1)
        if kv_proj is None and not (q is k and k is v):
            self._fuse()
            if k is v:
                kv_proj = self.project_kv(k)
            else:
                kk = ops.linear(k, self.w_ks.weight, self.w_ks.bias).reshape(-1, self.n_head * self.d_k)
                vv = ops.linear(v, self.w_vs.weight, self.w_vs.bias).reshape(-1, self.n_head * self.d_v)
                kv_proj = torch.cat([kk, vv], 1)
        out, attn = self.forward_rows(q.reshape(sz_b * len_q, d), sz_b, (len_q,), mask=mask, kv_proj=kv_proj)
        return out.view(sz_b, len_q, d), attn.view(self.n_head * sz_b, len_q, len_k)


class ScaledDotProductAttention(nn.Module):
    ''' Scaled Dot-Product Attention (attention.py:63-83).  q,k,v: (n_head*B, L, 64) as in the reference. '''

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)
        self.softmax = nn.Softmax(dim=2)

    def forward(self, q, k, v, mask=None):
        G, Lq, d = q.shape
        Lk = k.size(1)
        if d != 64:
            raise ValueError("the HIP attention kernel is built for head dim 64")
        mask_kind, mask_t = _as_mask(mask, G, Lq, Lk)
        drop_p = self.dropout.p if self.training else 0.0
        # every (group) row is its own single-head "batch" entry for the kernel
        out, attn = ops.SDPAFn.apply(q.contiguous(), k.contiguous(), v.contiguous(), 1, 1.0 / float(self.temperature),
                                     mask_kind, mask_t, drop_p)
        return out, attn
