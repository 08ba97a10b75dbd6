"""Mirror of SBL_Multilingual_Lip_reading/transformer/transformer.py."""
import torch.nn as nn

from .video_frontend import visual_frontend


class Transformer(nn.Module):
    """An encoder-decoder framework only includes attention (transformer.py:5-69)."""

    def __init__(self, encoder, decoder, pt):
        super(Transformer, self).__init__()
        self.visual_frontend = visual_frontend(pt)
        self.encoder = encoder
        self.decoder = decoder

        for p in self.parameters():          # transformer.py:18-20: overrides every sub-module init
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, padded_input, padded_target_l2r, padded_target_r2l):
        """
        Args:
            padded_input: N x T x H x W grayscale lip crops
            padded_targets: N x To (IGNORE_ID padded), l2r and r2l
        Returns: pred_l2r, gold_l2r, pred_r2l, gold_r2l
        """
        padded_input = padded_input.unsqueeze(4)                # gray channel (transformer.py:31)
        padded_input = padded_input.permute(0, 4, 1, 2, 3)      # (N,1,T,H,W); views only
        padded_input = self.visual_frontend(padded_input)
        batch = padded_input.size(0)
        input_lengths = [padded_input.size(1)] * batch          # transformer.py:37: always full length
        encoder_padded_outputs, *_ = self.encoder(padded_input, input_lengths)

        pred_l2r, gold_l2r, pred_r2l, gold_r2l = self.decoder(padded_target_l2r, padded_target_r2l,
                                                              encoder_padded_outputs, input_lengths)
        return pred_l2r, gold_l2r, pred_r2l, gold_r2l

    def recognize(self, input):
        """Greedy bidirectional decode (transformer.py:45-69).  input: N x T x H x W.
        Returns (ys_l2r, ys_r2l), int64 (N, 17)."""
        input = input.unsqueeze(4)
        input = input.permute(0, 4, 1, 2, 3)
        input = self.visual_frontend(input)
        batch = input.size(0)
        input_lengths = [input.size(1)] * batch
        encoder_outputs, *_ = self.encoder(input, input_lengths)
        ys_l2r, ys_r2l = self.decoder.recognize_beam(encoder_outputs)
        return ys_l2r, ys_r2l
