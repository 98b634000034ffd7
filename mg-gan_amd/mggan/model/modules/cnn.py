"""Scene CNN + physical attention with the reference's surface
(/root/reference/mggan/model/modules/cnn.py), computed by csrc/cnn.hip.  Only the
configuration instantiated on the hot path is supported (2 conv blocks, batch
norm, ReLU, no skip connections: standard.py:58-60, discriminators.py:71-73)."""
import numpy as np
import torch
import torch.nn as nn

from mggan.hip.flat import FlatModule
from mggan.hip import functions as HF


def attention_head(channels, hidden):
    """The channel-attention MLP of AttentionGlobal (reference: make_mlp([C, mlp_dim, C], ["leakyrelu", None]),
    cnn.py:102-107): Linear(C, hidden) -> LeakyReLU(0.01) -> Linear(hidden, C) as an nn.Sequential, so that the
    state_dict keys ("0.weight", "2.weight", ...) and the order of the seeded initialisation are the reference's."""
    return nn.Sequential(nn.Linear(channels, hidden), nn.LeakyReLU(), nn.Linear(hidden, channels))


class Conv_Blocks(nn.Module):
    def __init__(self, input_dim, output_dim, filter_size=3, batch_norm=False, non_lin="tanh", dropout=0.0,
                 first_block=False, last_block=False, skip_connection=False):
        super().__init__()
        if not batch_norm or non_lin != "relu" or dropout > 0 or skip_connection or filter_size != 3:
            raise ValueError("HIP Conv_Blocks supports 3x3 conv + BatchNorm + ReLU + MaxPool only")
        self.skip_connection = skip_connection
        self.last_block = last_block
        self.first_block = first_block
        self.Block = nn.Sequential()
        self.Block.add_module("Conv_1", nn.Conv2d(input_dim, output_dim, filter_size, 1, 1))
        self.Block.add_module("BN_1", nn.BatchNorm2d(output_dim))
        self.Block.add_module("NonLin_1", nn.ReLU())
        self.Block.add_module("Pool", nn.MaxPool2d(kernel_size=(2, 2), stride=(2, 2), dilation=(1, 1), ceil_mode=False))


class CNN(nn.Module):
    def __init__(self, social_pooling=False, channels_cnn=4, mlp=32, encoder_h_dim=16, insert_trajectory=False,
                 PhysFeature=False, margin_in=32, num_layers=3, dropout=0.0, batch_norm=False, non_lin_cnn="tanh",
                 in_channels=3, skip_connection=False):
        super().__init__()
        if num_layers != 2 or in_channels != 4 or insert_trajectory or social_pooling or channels_cnn not in (8, 16):
            raise ValueError("HIP CNN supports the hot-path configuration: 2 layers, 4 input channels, 8 or 16 filters")
        self.social_pooling = social_pooling
        self.in_traj = insert_trajectory
        self.skip_connection = skip_connection
        self.PhysFeature = PhysFeature
        self.bottleneck_dim = int(margin_in / 2 ** (num_layers - 1)) ** 2
        self.non_lin = non_lin_cnn
        self.encoder = nn.Sequential()
        self.encoder.add_module("ConvBlock_1", Conv_Blocks(in_channels, channels_cnn, dropout=dropout,
                                                           batch_norm=batch_norm, non_lin=self.non_lin,
                                                           first_block=True, skip_connection=skip_connection))
        self.encoder.add_module("ConvBlock_2", Conv_Blocks(channels_cnn, channels_cnn, dropout=dropout,
                                                           batch_norm=batch_norm, non_lin=self.non_lin,
                                                           skip_connection=skip_connection, last_block=True))
        self.bootleneck_channel = channels_cnn
        self.init_weights()

    def init_weights(self):
        def init_kaiming(m):
            if type(m) in [nn.Conv2d, nn.ConvTranspose2d]:
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_in")
                m.bias.data.fill_(0.01)

        self.apply(init_kaiming)


class AttentionGlobal(FlatModule):
    """AttentionGlobal(noise_attention_dim=0, PhysFeature=True, num_layers=2, channels_cnn=C).
    The reference constructs the CNN twice with the attention MLP in between
    (AttentionNetwork.__init__ then AttentionGlobal.__init__, cnn.py:56-107); the same
    order is kept so that seeded initialisation matches."""

    def __init__(self, noise_attention_dim=0, PhysFeature=True, num_layers=2, channels_cnn=4, mlp_dim=32, margin_in=16,
                 **kwargs):
        super().__init__()
        self.noise_attention_dim = noise_attention_dim
        self.mlp_dim = mlp_dim
        self.channels_cnn = channels_cnn
        self.num_layers = num_layers
        self.margin_in = margin_in
        self.PhysFeature = True
        self.skip_connection = False
        self.need_decoder = False
        self.sync = None  # set by mggan.parallel for multi-GPU batch statistics
        self.init_cnn()
        self.final_embedding = self.CNN.bottleneck_dim + self.noise_attention_dim
        self.cnn_attention = attention_head(self.CNN.bootleneck_channel, self.mlp_dim)
        self.init_cnn()

    def init_cnn(self):
        self.CNN = CNN(channels_cnn=self.channels_cnn, encoder_h_dim=128, mlp=self.mlp_dim, PhysFeature=True,
                       margin_in=self.margin_in, batch_norm=True, non_lin_cnn="relu", num_layers=self.num_layers,
                       in_channels=4)

    def forward(self, features, stat_updates=1, out=None):
        """features (B,4,33,33) -> (B,64).  stat_updates: how many reference forwards this call stands for
        (BatchNorm running statistics are updated that many times).  out: an HF.OutSlot to write the result into."""
        HF.root_of(self)
        b1, b2 = self.CNN.encoder.ConvBlock_1.Block, self.CNN.encoder.ConvBlock_2.Block
        a = self.cnn_attention
        return HF.SceneAttentionFn.apply(features, b1.Conv_1.weight, b1.Conv_1.bias, b1.BN_1.weight, b1.BN_1.bias,
                                         b2.Conv_1.weight, b2.Conv_1.bias, b2.BN_1.weight, b2.BN_1.bias, a[0].weight,
                                         a[0].bias, a[2].weight, a[2].bias, b1.BN_1, b2.BN_1, self.training, self,
                                         self.sync, HF.want_grad(a[0].weight), stat_updates, out)
