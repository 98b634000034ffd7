"""DiscreteLatentGenerator with the reference's class surface
(/root/reference/mggan/model/modules/standard_discrete.py:18-257, --experiment discrete): ONE decoder; the sampled
generator id goes through `one_hot_sample_encoder` and is appended to the encoder state before `enc_h_to_dec_h`."""
import numpy as np
import torch
import torch.nn as nn

from mggan.hip import functions as HF
from mggan.hip.flat import FlatModule
from mggan.model.modules.cnn import AttentionGlobal
from mggan.model.modules.common_modules import GeneratorOutput, RelativeDecoder, TrajectoryEncoder
from mggan.model.modules.social import SocialAttention
from mggan.model.modules.social_gan import PoolHiddenNet
from mggan.model.modules.standard import MultiGenerator
from mggan.rng import HostRNG
from mggan.utils import make_mlp


class DiscreteLatentGenerator(MultiGenerator):
    def __init__(self, z_size, encoder_h_dim, decoder_h_dim, social_feat_size, num_gens, pred_len, embedding_dim,
                 inp_format, num_social_modules, pool_type, scene_dim, use_pinet, learn_prior=False):
        FlatModule.__init__(self)
        assert inp_format in ("rel", "abs", "abs_rel")
        assert num_social_modules in (0, 1, num_gens)
        assert pool_type in ("sways", "sgan")
        if inp_format != "rel" or social_feat_size <= 0 or scene_dim <= 0 or social_feat_size != encoder_h_dim:
            raise ValueError("HIP DiscreteLatentGenerator implements the default configuration: inp_format='rel', "
                             "social and scene features enabled")
        self.use_pinet, self.inp_format, self.z_size = use_pinet, inp_format, z_size
        self.embedding_dim, self.social_feat_size = embedding_dim, social_feat_size
        self.n_social_modules, self.pool_type = num_social_modules, pool_type
        self.decoder_h_dim, self.encoder_h_dim, self.scene_dim = decoder_h_dim, encoder_h_dim, scene_dim
        # (module construction order = the reference's: seeded initialisation is bit-identical)
        self.encoder = TrajectoryEncoder(inp_size=2, hidden_size=encoder_h_dim, embedding_dim=embedding_dim, num_layers=1)
        self.scene_encoder = AttentionGlobal(noise_attention_dim=0, PhysFeature=True, num_layers=2, channels_cnn=16)
        if pool_type == "sways":
            self.social = SocialAttention(social_feat_size, encoder_h_dim)
        else:
            self.social = PoolHiddenNet(embedding_dim=embedding_dim, h_dim=encoder_h_dim, mlp_dim=social_feat_size,
                                        bottleneck_dim=encoder_h_dim)
        self.decoder = RelativeDecoder(pred_len=pred_len, embedding_dim=embedding_dim, h_dim=decoder_h_dim, num_layers=1,
                                       social_feat_size=encoder_h_dim, z_size=z_size, dropout=0.0, inp_format=inp_format)
        self.n_gs = num_gens
        self.pred_len = pred_len
        self.enc_h_to_dec_h = make_mlp([encoder_h_dim + z_size + scene_dim + z_size + social_feat_size, decoder_h_dim],
                                       batch_norm=False)
        assert not (use_pinet and learn_prior), "Using conditional distribution already, `learn_prior` has no effect"
        self.net_chooser = nn.Sequential(
            nn.Linear(encoder_h_dim + scene_dim + social_feat_size, encoder_h_dim // 2), nn.ReLU(),
            nn.Linear(encoder_h_dim // 2, encoder_h_dim // 2), nn.ReLU(),
            nn.Linear(encoder_h_dim // 2, num_gens))
        self.one_hot_sample_encoder = make_mlp([num_gens, z_size, z_size])
        self.net_prior = nn.Parameter(torch.zeros(1, self.n_gs), requires_grad=learn_prior)
        self.rng = HostRNG()

    def generator_parameters(self):
        return list(self.decoder.parameters())

    def _rows(self, R, dev):
        """Every row its own 'pedestrian', one generator, noise slot 0."""
        cache = self.__dict__.setdefault("_rows_cache", HF.BoundedCache(16))
        key = (R, str(dev))
        rows = cache.get(key)
        if rows is None:
            rows = cache.put(key, HF.RolloutRows(np.zeros(R, np.int64), np.arange(R), np.zeros(R, np.int64), 1, R, dev))
        return rows

    def _code(self, one_hot_rows):
        e = self.one_hot_sample_encoder
        return HF.mlp(one_hot_rows, [(e[0], HF.ACT_LEAKY, 0.0), (e[2], HF.ACT_NONE, 0.0)])

    def _roll(self, in_xy, in_dxdy, enc_rows, soc_rows, noise_rows, reps):
        """One decoder over R = reps*b rows (standard_discrete.py:236-257)."""
        R = enc_rows.shape[0]
        e2d = self.enc_h_to_dec_h[0]
        w_hh = self.decoder.decoder.weight_hh_l0
        return HF.DecoderRolloutFn.apply(enc_rows, soc_rows, noise_rows.reshape(1, R, -1), in_xy[-1].repeat(reps, 1),
                                         in_dxdy[-1].repeat(reps, 1), self._rows(R, enc_rows.device), e2d.weight,
                                         e2d.bias, w_hh, self.decoder.param_dict(), 1, 0, self.pred_len, self,
                                         HF.want_grad(enc_rows, soc_rows, w_hh))

    def forward_all(self, in_xy, in_dxdy, enc_h, noise, social_feats):
        """The decoder on the current batch (standard_discrete.py:236-257): enc_h (b, enc + z) already carries the
        embedded generator id, noise (b, z) -> (abs, rel), each (pred_len, b, 2)."""
        return self._roll(in_xy, in_dxdy, enc_h, social_feats, noise, 1)

    def forward(self, in_xy, in_dxdy, sub_batches, noise=None, all_gen_out=True, img=None, num_samples=5, mask=None,
                trunk=None, logits=None, need_samples=True):
        """Returns (GeneratorOutput(rel, abs), net_chooser_out (b_m, g), sampled_gen_idxs (b_m, K) int64)."""
        if img is None:
            raise ValueError("img is mandatory: scene_dim=64 is hard-wired into the model (SURVEY A.6)")
        self.ensure_flat()
        batch_size = in_xy.size(1)
        dev = in_xy.device
        enc_h, soc = trunk if trunk is not None else self.trunk(in_xy, in_dxdy, sub_batches, img)
        if noise is not None:
            assert noise.shape == (num_samples, batch_size, self.z_size)
        else:
            noise = self.rng.noise(num_samples, self.z_size, sub_batches, dev)
        if mask is not None and not bool(mask.all()):
            in_xy, in_dxdy = in_xy[:, mask], in_dxdy[:, mask]
            enc_h, soc, noise = enc_h[mask], soc[mask], noise[:, mask]
            batch_size = int(mask.sum())
        b, g, K = batch_size, self.n_gs, num_samples
        eye = torch.eye(g, device=dev)
        if all_gen_out:
            with torch.no_grad():  # rows ((k*g + j)*b + ped): generator j on sample k of pedestrian ped
                code = self._code(eye).repeat_interleave(b, 0).repeat(K, 1)
                enc_rows = torch.cat([enc_h.detach().repeat(K * g, 1), code], 1)
                nz = noise[:, None].expand(K, g, b, self.z_size).reshape(K * g * b, -1)
                pa, pr = self._roll(in_xy, in_dxdy, enc_rows, soc.detach().repeat(K * g, 1), nz, K * g)
            if need_samples or not getattr(self.rng, "on_device", False):
                net_chooser_out, sampled = self.get_samples(enc_h, num_samples)
            else:
                net_chooser_out, sampled = self._chooser(enc_h) if self.use_pinet else \
                    self.net_prior.expand(enc_h.size(0), -1), None
            shape = (self.pred_len, K, g, b, 2)
            return GeneratorOutput(pr.view(shape), pa.view(shape)), net_chooser_out, sampled
        with torch.no_grad():
            net_chooser_out, sampled = self.get_samples(enc_h, num_samples, logits=logits)
        gen_rows = sampled.t().reshape(-1)                       # rows k*b + ped
        if sampled.is_cuda and getattr(self.rng, "on_device", False):
            self.last_rows = HF.device_rollout_rows(sampled, g)  # (the trainer reads the row ids / counts from here)
        else:
            self.last_rows = None
        code = self._code(eye[gen_rows.to(dev)])                 # sample k uses noise[k] (no occurrence-offset slots)
        enc_rows = torch.cat([enc_h.repeat(K, 1), code], 1)
        pa, pr = self._roll(in_xy, in_dxdy, enc_rows, soc.repeat(K, 1), noise.reshape(K * b, -1), K)
        shape = (self.pred_len, K, b, 2)
        return GeneratorOutput(pr.view(shape), pa.view(shape)), net_chooser_out, sampled
