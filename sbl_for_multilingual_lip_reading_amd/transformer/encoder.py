"""Mirror of SBL_Multilingual_Lip_reading/transformer/encoder.py."""
import torch
import torch.nn as nn

from ._env import ops
from .attention import MultiHeadAttention
from .module import PositionalEncoding, PositionwiseFeedForward
from .utils import get_non_pad_mask, get_attn_pad_mask


class Encoder(nn.Module):
    """Encoder of Transformer including self-attention and feed forward (encoder.py:8-67)."""

    def __init__(self, d_input, n_layers, n_head, d_k, d_v,
                 d_model, d_inner, dropout=0.1, pe_maxlen=5000):
        super(Encoder, self).__init__()
        self.d_input = d_input
        self.n_layers = n_layers
        self.n_head = n_head
        self.d_k = d_k
        self.d_v = d_v
        self.d_model = d_model
        self.d_inner = d_inner
        self.dropout_rate = dropout
        self.pe_maxlen = pe_maxlen

        self.linear_in = nn.Linear(d_input, d_model)
        self.layer_norm_in = nn.LayerNorm(d_model)
        self.positional_encoding = PositionalEncoding(d_model, max_len=pe_maxlen)
        self.dropout = nn.Dropout(dropout)

        self.layer_stack = nn.ModuleList([
            EncoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout)
            for _ in range(n_layers)])

    def forward(self, padded_input, input_lengths, return_attns=False):
        """
        Args:
            padded_input: N x T x D
            input_lengths: N
        Returns:
            enc_output: N x T x H
        """
        enc_slf_attn_list = []
        N, T, _ = padded_input.shape
        full = all(int(l) >= T for l in input_lengths)
        if full:
            # every mask of encoder.py:47-49 is all-ones / all-false (transformer.py:37): skip them
            non_pad_mask, slf_attn_mask = None, None
        else:
            non_pad_mask = get_non_pad_mask(padded_input, input_lengths=input_lengths)
            slf_attn_mask = get_attn_pad_mask(padded_input, input_lengths, T)

        # the layers' weight gradients are deferred and issued as one grouped launch when backward reaches the
        # encoder input (second stream, under the frontend backward) or at the end of backward, whichever is first
        defer = torch.is_grad_enabled()
        if defer:
            ops.begin_defer()
            if padded_input.requires_grad:
                padded_input.register_hook(lambda g: ops.flush_deferred())
        # dropout(LayerNorm(linear_in(x)) + pe[:T])  (encoder.py:53-55)
        h = ops.linear(padded_input, self.linear_in.weight, self.linear_in.bias)
        h = ops.add_layernorm(h, None, self.layer_norm_in.weight, self.layer_norm_in.bias, self.layer_norm_in.eps)
        h = ops.AddPEFn.apply(h, self.positional_encoding.pe[0, :T])
        enc_output = ops.dropout(h, self.dropout.p, self.training)

        for enc_layer in self.layer_stack:
            enc_output, enc_slf_attn = enc_layer(
                enc_output,
                non_pad_mask=non_pad_mask,
                slf_attn_mask=slf_attn_mask)
            if return_attns:
                enc_slf_attn_list += [enc_slf_attn]

        if defer:
            ops.end_defer()
        if return_attns:
            return enc_output, enc_slf_attn_list
        return enc_output,


class EncoderLayer(nn.Module):
    """Compose with two sub-layers (encoder.py:70-91)."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super(EncoderLayer, self).__init__()
        self.slf_attn = MultiHeadAttention(
            n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(
            d_model, d_inner, dropout=dropout)

    def forward(self, enc_input, non_pad_mask=None, slf_attn_mask=None):
        enc_output, enc_slf_attn = self.slf_attn(
            enc_input, enc_input, enc_input, mask=slf_attn_mask)
        if non_pad_mask is not None:
            enc_output = ops.RowScaleFn.apply(enc_output, non_pad_mask)

        enc_output = self.pos_ffn(enc_output)
        if non_pad_mask is not None:
            enc_output = ops.RowScaleFn.apply(enc_output, non_pad_mask)

        return enc_output, enc_slf_attn
