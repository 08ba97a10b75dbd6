"""Mirror of SBL_Multilingual_Lip_reading/transformer/module.py."""
import math

import torch
import torch.nn as nn

from ._env import ops


class PositionalEncoding(nn.Module):
    """PE(pos, 2i) = sin(pos/10000^(2i/d)), PE(pos, 2i+1) = cos(...) — module.py:8-32.  The table is a buffer
    named 'pe' of shape (1, max_len, d_model), built in log space in fp32 exactly like the reference (host-side,
    construction time only)."""

    def __init__(self, d_model, max_len=5000):
        super(PositionalEncoding, self).__init__()
        pe = torch.zeros(max_len, d_model, requires_grad=False)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        pe = pe.unsqueeze(0)
        self.register_buffer('pe', pe)

    def forward(self, input):
        length = input.size(1)
        return self.pe[:, :length]


class PositionwiseFeedForward(nn.Module):
    """FFN(x) = LayerNorm(dropout(max(0, xW1 + b1)W2 + b2) + x) — module.py:35-52, one fused tape node."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, x):
        drop_p = self.dropout.p if self.training else 0.0
        return ops.FFNFn.apply(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                               self.layer_norm.weight, self.layer_norm.bias, drop_p, self.layer_norm.eps)
