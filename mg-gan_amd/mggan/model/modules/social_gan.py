"""PoolHiddenNet with the reference's class surface
(/root/reference/mggan/model/modules/social_gan.py:157-229), used for --pool_type sgan in G and D."""
import torch.nn as nn

from mggan.hip import functions as HF
from mggan.hip.flat import FlatModule


class PoolHiddenNet(FlatModule):
    """Pooling module of Social GAN: per scene, every pedestrian pools (max) an MLP of [embedded relative position |
    neighbour hidden state] over all pedestrians of the scene, itself included."""

    def __init__(self, embedding_dim=64, h_dim=64, mlp_dim=1024, bottleneck_dim=1024, activation="relu", batch_norm=False,
                 dropout=0.0):
        super().__init__()
        if activation != "relu" or batch_norm or dropout:
            raise ValueError("HIP path implements PoolHiddenNet as the model factory builds it: ReLU, no BatchNorm, "
                             "no dropout")
        self.mlp_dim, self.h_dim, self.bottleneck_dim, self.embedding_dim = mlp_dim, h_dim, bottleneck_dim, embedding_dim
        self.spatial_embedding = nn.Linear(2, embedding_dim)
        # utils.make_mlp([E + H, H, bottleneck]): Linear, ReLU, Linear (no activation after the last layer)
        self.mlp_pre_pool = nn.Sequential(nn.Linear(embedding_dim + h_dim, h_dim), nn.ReLU(), nn.Linear(h_dim, bottleneck_dim))

    def forward(self, in_xy, in_dxdy, h_states, seq_start_end, xy_mod=0):
        """in_xy (T,N,2), h_states (N,h) -> (sum of the list's scene sizes, bottleneck), rows in LIST order
        (a list repeated K times yields K copies, social_gan.py:212-228)."""
        HF.root_of(self)
        tb = HF.pool_tables(seq_start_end, h_states.device)
        m = self.mlp_pre_pool
        h = h_states.reshape(-1, self.h_dim)
        return HF.PoolHiddenFn.apply(in_xy[-1], h, tb, self.spatial_embedding.weight, self.spatial_embedding.bias,
                                     m[0].weight, m[0].bias, m[2].weight, m[2].bias, self,
                                     HF.want_grad(h_states, m[0].weight), xy_mod)
