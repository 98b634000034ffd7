"""MultiGenerator with the reference's class surface
(/root/reference/mggan/model/modules/standard.py:17-265): trajectory encoder + scene
attention + social attention trunk, `num_gens` LSTM decoders, PM-network (`net_chooser`).

MI355X design differences (results identical): instead of running every
generator for `max occurrence` noise slots and gathering (standard.py:190-214), exactly
one rollout per selected (pedestrian, sample) is launched, rows bucketed by generator."""
import numpy as np
import torch
import torch.nn as nn

from mggan.utils import make_mlp, get_selection_indices
from mggan.rng import HostRNG
from mggan.hip.flat import FlatModule
from mggan.hip import functions as HF
from mggan.model.modules.cnn import AttentionGlobal
from mggan.model.modules.social_gan import PoolHiddenNet
from mggan.model.modules.social import SocialAttention
from mggan.model.modules.common_modules import TrajectoryEncoder, RelativeDecoder, get_input, GeneratorOutput


class MultiGenerator(FlatModule):
    def __init__(self, z_size, encoder_h_dim, decoder_h_dim, social_feat_size, num_gens, pred_len, embedding_dim,
                 inp_format, num_social_modules, pool_type, scene_dim, use_pinet, learn_prior=False):
        super().__init__()
        assert inp_format in ("rel", "abs", "abs_rel")
        assert num_social_modules in (0, 1, num_gens)
        assert pool_type in ("sways", "sgan")
        if inp_format != "rel" or social_feat_size <= 0 or scene_dim <= 0:
            raise ValueError("HIP MultiGenerator implements the default hot path: inp_format='rel', "
                             "social and scene features enabled")
        if pool_type == "sgan" and social_feat_size != encoder_h_dim:
            raise ValueError("pool_type 'sgan' pools to encoder_h_dim features; social_feat_size must equal it")
        self.use_pinet = use_pinet
        self.inp_format = inp_format
        self.z_size = z_size
        self.embedding_dim = embedding_dim
        self.social_feat_size = social_feat_size
        self.n_social_modules = num_social_modules
        self.pool_type = pool_type
        self.decoder_h_dim = decoder_h_dim
        self.encoder_h_dim = encoder_h_dim
        self.scene_dim = scene_dim

        self.encoder = TrajectoryEncoder(inp_size=2, hidden_size=encoder_h_dim, embedding_dim=embedding_dim,
                                         num_layers=1)
        self.scene_encoder = AttentionGlobal(noise_attention_dim=0, PhysFeature=True, num_layers=2, channels_cnn=16)
        if pool_type == "sways":
            self.social = SocialAttention(social_feat_size, encoder_h_dim)
        else:  # standard.py:66-71
            self.social = PoolHiddenNet(embedding_dim=embedding_dim, h_dim=encoder_h_dim, mlp_dim=social_feat_size,
                                        bottleneck_dim=encoder_h_dim)

        self.gs = nn.ModuleList()
        for i in range(num_gens):
            decoder = RelativeDecoder(pred_len=pred_len, embedding_dim=embedding_dim, h_dim=decoder_h_dim,
                                      num_layers=1, social_feat_size=social_feat_size, z_size=z_size, dropout=0.0,
                                      inp_format=inp_format)
            setattr(self, "G_{}".format(i), decoder)
            self.gs.append(decoder)

        self.n_gs = len(self.gs)
        self.pred_len = pred_len
        self.enc_h_to_dec_h = make_mlp([encoder_h_dim + z_size + scene_dim + social_feat_size, decoder_h_dim],
                                       batch_norm=False)
        assert not (use_pinet and learn_prior), "Using conditional distribution already, `learn_prior` has no effect"
        self.net_chooser = nn.Sequential(
            nn.Linear(encoder_h_dim + scene_dim + social_feat_size, encoder_h_dim // 2), nn.ReLU(),
            nn.Linear(encoder_h_dim // 2, encoder_h_dim // 2), nn.ReLU(),
            nn.Linear(encoder_h_dim // 2, num_gens))
        self.net_prior = nn.Parameter(torch.zeros(1, self.n_gs), requires_grad=learn_prior)
        self.rng = HostRNG()

    # -- helpers ---------------------------------------------------------------------------
    def generator_parameters(self):
        hit = self.__dict__.get("_gen_params")
        if hit is None or hit[0] is not self._flat or self._flat is None:
            hit = (self._flat, [p for g in self.gs for p in g.parameters()])
            self.__dict__["_gen_params"] = hit
        return hit[1]

    def _gen_stride(self):
        if self.n_gs == 1:
            return 0
        d0, d1 = self.gs[0].param_dict(), self.gs[1].param_dict()
        strides = {(d1[k].data_ptr() - d0[k].data_ptr()) // 4 for k in d0}
        assert len(strides) == 1, "per-generator parameter blocks must be laid out with a constant stride"
        return strides.pop()

    def _all_rows(self, n, b, dev):
        """Row table of "every generator on every (sample, pedestrian)": static per shape, cached."""
        key = (n, b, str(dev))
        cache = self.__dict__.setdefault("_all_rows_cache", HF.BoundedCache(16))
        rows = cache.get(key)
        if rows is None:
            g = self.n_gs
            rows = cache.put(key, HF.RolloutRows(np.tile(np.repeat(np.arange(g), b), n), np.tile(np.arange(b), n * g),
                                                 np.repeat(np.arange(n), g * b), g, b, dev))
        return rows

    def trunk(self, in_xy, in_dxdy, sub_batches, img, passes=1):
        """-> (enc_h (b,128) = [lstm | scene | social], social (b,32)).  `passes` > 1: the result stands
        for that many identical reference forwards (the no-grad generator call of the discriminator step and
        the generator step see the same weights) -> BatchNorm running stats are updated that many times."""
        self.ensure_flat()
        b = in_xy.size(1)
        if self.pool_type != "sways" or b == 0:
            with HF.branch():  # scene CNN || trajectory LSTM + social pooling
                scene = self.scene_encoder(img, stat_updates=passes)
            enc = self.encoder(get_input(in_xy, in_dxdy, self.inp_format))
            soc = self.social(in_xy, in_dxdy, enc, sub_batches)
            HF.join_branch(scene)
            return torch.cat([enc, scene, soc], -1), soc
        # enc_h = [lstm | scene | social] is ONE (b, 128) buffer: the three producers write their column blocks in
        # place (HF.OutSlot), HF.TrunkJoinFn runs the social attention and hands out enc_h -- no torch.cat
        He, Sc, Fs = self.encoder_h_dim, self.scene_dim, self.social_feat_size
        buf = torch.empty(b, He + Sc + Fs, dtype=torch.float32, device=in_xy.device)
        with HF.branch():  # scene CNN || trajectory LSTM + social attention
            HF.mark("trunk.cnn.begin")
            scene = self.scene_encoder(img, stat_updates=passes, out=HF.OutSlot(buf, He, He + Sc))
            HF.mark("trunk.cnn.end")
        enc = self.encoder(get_input(in_xy, in_dxdy, self.inp_format), out=HF.OutSlot(buf, 0, He))
        tb = HF.scene_tables(sub_batches, b, in_xy.device)
        fc, W = self.social.feature_embedder.fc, self.social.attention.W
        enc_h, soc = HF.TrunkJoinFn.apply(enc, in_xy[-1], in_dxdy[-1], tb, buf, fc[0].weight, fc[0].bias, fc[2].weight,
                                          fc[2].bias, fc[4].weight, fc[4].bias, W.weight, W.bias, self.social,
                                          HF.want_grad(enc, W.weight), Sc)
        HF.mark("trunk.soc.end")
        HF.join_branch(scene)  # the scene block of enc_h is complete from here on
        enc_h = HF.ColsTapFn.apply(enc_h, scene, He, He + Sc)
        return enc_h, soc

    def _chooser(self, enc_h):
        nc = self.net_chooser
        return HF.mlp(enc_h, [(nc[0], HF.ACT_LEAKY, 0.0), (nc[2], HF.ACT_LEAKY, 0.0), (nc[4], HF.ACT_NONE, 0.0)])

    def _rollout(self, rows, in_xy, in_dxdy, enc_h, social_feats, noise):
        e2d = self.enc_h_to_dec_h[0]
        w_hh = self.gs[0].decoder.weight_hh_l0
        return HF.DecoderRolloutFn.apply(enc_h, social_feats, noise, in_xy[-1], in_dxdy[-1], rows, e2d.weight, e2d.bias,
                                         w_hh, self.gs[0].param_dict(), self.n_gs, self._gen_stride(), self.pred_len,
                                         self, HF.want_grad(enc_h, social_feats, w_hh))

    # -- reference surface -------------------------------------------------------------------
    def forward(self, in_xy, in_dxdy, sub_batches, noise=None, all_gen_out=True, img=None, num_samples=5, mask=None,
                trunk=None, logits=None, need_samples=True):
        """Returns (GeneratorOutput(rel, abs), net_chooser_out (b_m, g), sampled_gen_idxs (b_m, K) int64).
        abs/rel: (pred_len, K, b_m, 2) or (pred_len, K, g, b_m, 2) when all_gen_out."""
        if img is None:
            raise ValueError("img is mandatory: scene_dim=64 is hard-wired into the model (SURVEY A.6)")
        self.ensure_flat()
        batch_size = in_xy.size(1)
        dev = in_xy.device
        enc_h, social_feats = trunk if trunk is not None else self.trunk(in_xy, in_dxdy, sub_batches, img)

        if noise is not None:
            assert noise.shape == (num_samples, batch_size, self.z_size)
            noise = noise.to(dev)
        else:
            noise = self.rng.noise(num_samples, self.z_size, sub_batches, dev)

        if mask is not None and not bool(mask.all()):  # pass mask=None (all valid) to avoid this device sync
            in_xy, in_dxdy = in_xy[:, mask], in_dxdy[:, mask]
            enc_h, social_feats, noise = enc_h[mask], social_feats[mask], noise[:, mask]
            batch_size = int(mask.sum())
        b, g, K = batch_size, self.n_gs, num_samples
        ped = np.tile(np.arange(b), K)

        if all_gen_out:
            with torch.no_grad():
                pa, pr = self._rollout(self._all_rows(K, b, dev), in_xy, in_dxdy, enc_h.detach(),
                                       social_feats.detach(), noise)
            if need_samples or not getattr(self.rng, "on_device", False):
                net_chooser_out, sampled_gen_idxs = self.get_samples(enc_h, num_samples)
            else:  # caller ignores the draw and the device generator owes nobody a seed-compatible stream
                net_chooser_out, sampled_gen_idxs = self._chooser(enc_h) if self.use_pinet else \
                    self.net_prior.expand(enc_h.size(0), -1), None
            shape = (self.pred_len, K, g, b, 2)
            return GeneratorOutput(pr.view(shape), pa.view(shape)), net_chooser_out, sampled_gen_idxs

        rows = None
        with torch.no_grad():  # `logits`: the PM-network output of the same trunk and weights, computed by an earlier call
            if getattr(self.rng, "on_device", False) and hasattr(self.rng, "sample_rows") and enc_h.is_cuda:
                net_chooser_out = logits if logits is not None else self._chooser(enc_h) if self.use_pinet else \
                    self.net_prior.expand(enc_h.size(0), -1)
                sampled_gen_idxs, rows = self.rng.sample_rows(net_chooser_out, num_samples)  # picks + row tables, one launch
                # (a caller's hook: work that should start behind the sampling launches but beside the rollouts, e.g. the
                #  discriminator's scene CNN in the generator step -- its persistent grids, started first, keep the few
                #  workgroups of the sampling kernels waiting for a CU slot)
                cb, self._after_sampling = getattr(self, "_after_sampling", None), None
                if cb is not None:
                    cb()
            else:
                net_chooser_out, sampled_gen_idxs = self.get_samples(enc_h, num_samples, logits=logits)
        if rows is not None:
            pass
        elif sampled_gen_idxs.is_cuda and getattr(self.rng, "on_device", False):
            rows = HF.device_rollout_rows(sampled_gen_idxs, g)  # no host round trip
        else:
            idx_host = sampled_gen_idxs.cpu()
            offsets = get_selection_indices(idx_host)  # noise slot = occurrence offset, NOT the sample index (A.4)
            rows = HF.RolloutRows(idx_host.t().reshape(-1).numpy(), ped, offsets.t().reshape(-1).numpy(), g, b, dev)
        self.last_rows = rows
        pa, pr = self._rollout(rows, in_xy, in_dxdy, enc_h, social_feats, noise)
        shape = (self.pred_len, K, b, 2)
        return GeneratorOutput(pr.view(shape), pa.view(shape)), net_chooser_out, sampled_gen_idxs

    def get_samples(self, enc_h, num_samples=5, logits=None):
        """Returns (logits (b, g), generator indexes (b, num_samples))."""
        if logits is not None:
            net_chooser_out = logits
        elif self.use_pinet:
            net_chooser_out = self._chooser(enc_h)
        else:
            net_chooser_out = self.net_prior.expand(enc_h.size(0), -1)
        sampled = self.rng.sample_generators(net_chooser_out, num_samples).to(enc_h.device)
        return net_chooser_out, sampled

    def forward_all(self, in_xy, in_dxdy, enc_h, noise, social_feats):
        """Every generator on every (sample, pedestrian): two tensors (pred_len, n_samples, num_gens, b, 2)."""
        n, b, _ = noise.shape
        g = self.n_gs
        pa, pr = self._rollout(self._all_rows(n, b, enc_h.device), in_xy, in_dxdy, enc_h, social_feats, noise)
        shape = (self.pred_len, n, g, b, 2)
        return pa.view(shape), pr.view(shape)
