"""TrajectoryEncoder / RelativeDecoder with the reference's surface
(/root/reference/mggan/model/modules/common_modules.py), computed by the HIP
rollout kernels (csrc/lstm.hip)."""
from collections import namedtuple

import torch
from torch import nn

from mggan.utils import make_mlp
from mggan.hip.flat import FlatModule
from mggan.hip import functions as HF
from mggan.model.widths import holders_only

GeneratorOutput = namedtuple("generator_out", ["rel", "abs"])


def get_input(xy, dxdy, inp_format):
    if inp_format == "rel":
        inp = dxdy
    elif inp_format == "abs":
        inp = xy
    else:
        if xy.size(0) == (dxdy.size(0) + 1):
            dxdy = torch.cat([dxdy[0:1], dxdy], 0)
        inp = torch.cat([xy, dxdy], dim=2)
    return inp


class TrajectoryEncoder(FlatModule):
    def __init__(self, hidden_size=128, inp_size=2, num_layers=1, embedding_dim=None, return_hc=False):
        super().__init__()
        if inp_size != 2 or num_layers != 1 or embedding_dim is None or return_hc:
            raise ValueError("HIP TrajectoryEncoder supports inp_size=2 (inp_format 'rel'/'abs'), one layer, "
                             "an embedding and return_hc=False (the configurations of the reference hot path)")
        if hidden_size not in (32, 64) and not holders_only():
            raise ValueError("HIP TrajectoryEncoder: hidden_size {} not built (32: generator, 64: discriminator; "
                             "csrc/lstm.hip lstm_*_kernel<H>)".format(hidden_size))
        self.embedding_dim = embedding_dim
        self.inp_size = inp_size
        self.return_hc = return_hc
        self.embedding = nn.Linear(inp_size, embedding_dim)
        self.encoder = nn.LSTM(input_size=embedding_dim, hidden_size=hidden_size, num_layers=num_layers)

    def forward(self, inp, hc=None, out=None):
        """inp (T, b, 2) -> h_T (b, hidden).  out: an HF.OutSlot -- the result is written into (and returned as) that
        column block of a wider buffer instead of a fresh tensor."""
        if hc is not None:
            raise ValueError("initial state is always zero on the reference hot path")
        L = self.encoder
        HF.root_of(self)
        return HF.LstmEncoderFn.apply(inp, self.embedding.weight, self.embedding.bias, L.weight_ih_l0, L.weight_hh_l0,
                                      L.bias_ih_l0, L.bias_hh_l0, self, HF.want_grad(L.weight_hh_l0), out)


class RelativeDecoder(FlatModule):
    def __init__(self, pred_len=12, embedding_dim=128, h_dim=128, num_layers=1, dropout=0.0, inp_format="abs_rel",
                 z_size=64, social_feat_size=128):
        super().__init__()
        if inp_format != "rel" or num_layers != 1 or dropout != 0.0:
            raise ValueError("HIP RelativeDecoder supports inp_format='rel', one layer, no dropout")
        if (h_dim != 32 or not 0 < social_feat_size <= 32 or z_size < 4 or z_size % 4) and not holders_only():
            raise ValueError("HIP RelativeDecoder: h_dim {} / social_feat_size {} / z_size {} not built (32, 1..32, a "
                             "multiple of 4; csrc/lstm.hip decoder_*_kernel)".format(h_dim, social_feat_size, z_size))
        self.pred_len = pred_len
        self.h_dim = h_dim
        self.embedding_dim = embedding_dim
        self.inp_format = inp_format
        self.decoder = nn.LSTM(embedding_dim, h_dim, num_layers, dropout=dropout)
        self.spatial_embedding = nn.Linear(2, embedding_dim)
        self.hidden2pos = make_mlp([h_dim + social_feat_size, h_dim // 2, 2], "leaky_relu", batch_norm=False)

    def param_dict(self):
        L = self.decoder
        return {"emb_w": self.spatial_embedding.weight, "emb_b": self.spatial_embedding.bias, "w_ih": L.weight_ih_l0,
                "w_hh": L.weight_hh_l0, "b_ih": L.bias_ih_l0, "b_hh": L.bias_hh_l0, "w1": self.hidden2pos[0].weight,
                "b1": self.hidden2pos[0].bias, "w2": self.hidden2pos[2].weight, "b2": self.hidden2pos[2].bias}

    def generator_parameters(self):
        return list(self.parameters())

    def forward(self, xy, dxdy, noise, social_feats, state_tuple):
        """Stand-alone rollout (reference signature): returns (abs (T,R,2), rel (T,R,2)).
        h0 is taken from state_tuple; c0 must be zero as on the reference path."""
        root = HF.root_of(self)
        h0 = state_tuple[0][-1]
        R, H = h0.shape
        dev = h0.device
        eye = torch.eye(H, device=dev)
        zb = torch.zeros(H, device=dev)
        ar = torch.arange(R)
        rows = HF.RolloutRows(torch.zeros(R, dtype=torch.long), ar, torch.zeros(R, dtype=torch.long), 1, R, dev)
        noise0 = torch.zeros(1, R, 0, device=dev)
        save = HF.want_grad(h0, social_feats, self.decoder.weight_hh_l0)
        return HF.DecoderRolloutFn.apply(h0, social_feats, noise0, xy, dxdy, rows, eye, zb, self.decoder.weight_hh_l0,
                                         self.param_dict(), 1, 0, self.pred_len, self, save)
