"""Social attention (Social-Ways style) with the reference's surface
(/root/reference/mggan/model/modules/social.py), computed over in-scene pairs
only by csrc/social.hip."""
import torch
from torch import nn

from mggan.hip.flat import FlatModule
from mggan.hip import functions as HF
from mggan.model.widths import holders_only


class AttentionPooling(nn.Module):
    def __init__(self, h_dim, f_dim):
        super().__init__()
        self.f_dim = f_dim
        self.h_dim = h_dim
        self.W = nn.Linear(h_dim, f_dim, bias=True)


class EmbedSocialFeatures(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.fc = nn.Sequential(nn.Linear(input_size, 32), nn.ReLU(), nn.Linear(32, 64), nn.ReLU(),
                                nn.Linear(64, hidden_size))


class SocialAttention(FlatModule):
    def __init__(self, social_feat_size, hidden_size):
        super().__init__()
        if (hidden_size not in (32, 64) or social_feat_size < 1) and not holders_only():
            raise ValueError("HIP SocialAttention: hidden_size {} not built (32 or 64; csrc/social_rows.hip "
                             "social_rows_*_kernel<H>, csrc/social.hip)".format(hidden_size))
        self.feature_embedder = EmbedSocialFeatures(3, social_feat_size)
        self.attention = AttentionPooling(hidden_size, social_feat_size)

    def forward(self, in_xy, in_dxdy, enc_h, sub_batches, xy_mod=0):
        """in_xy (T,N,2), in_dxdy (T-1,N,2), enc_h (N,h) -> (N,h).  Only rows covered by
        sub_batches receive features (the discriminator passes a list repeated K times that
        still indexes the first b rows).  xy_mod > 0: enc_h / sub_batches cover several
        repetitions of the xy_mod pedestrians that in_xy / in_dxdy describe (pair pass)."""
        HF.root_of(self)
        fc, W = self.feature_embedder.fc, self.attention.W
        b = max(int(e) for _, e in sub_batches) if len(sub_batches) else 0
        N = enc_h.shape[0]
        tb = HF.scene_tables(sub_batches, b, enc_h.device)
        xy_l, dxy_l, h = in_xy[-1], in_dxdy[-1], enc_h
        if N != b:  # (identity slices would still cost a zero-fill + copy each in autograd's slice backward)
            h = h[:b]
            if not xy_mod:
                xy_l, dxy_l = xy_l[:b], dxy_l[:b]
        S = HF.SocialAttentionFn.apply(xy_l, dxy_l, h, tb, fc[0].weight, fc[0].bias,
                                       fc[2].weight, fc[2].bias, fc[4].weight, fc[4].bias, W.weight, W.bias, self,
                                       HF.want_grad(enc_h, W.weight), xy_mod)
        if N > b:
            S = torch.cat([S, S.new_zeros(N - b, S.shape[1])], 0)
        return S
