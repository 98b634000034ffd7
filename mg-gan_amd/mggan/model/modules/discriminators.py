"""MultiDiscriminatorTrajectory with the reference's class surface
(/root/reference/mggan/model/modules/discriminators.py:12-219), gan_type 'mgan' / 'gan'."""
import contextlib

import torch
import torch.nn as nn

from mggan.hip.flat import FlatModule
from mggan.hip import functions as HF
from mggan.model.modules.cnn import AttentionGlobal
from mggan.model.modules.social import SocialAttention
from mggan.model.modules.social_gan import PoolHiddenNet
from mggan.model.modules.common_modules import TrajectoryEncoder


class MultiDiscriminatorTrajectory(FlatModule):
    def __init__(self, num_gens, num_discs, unbound_output, h_dim, inp_format, pred_len, gan_type, global_disc,
                 scene_dim, pool_type="sgan"):
        super().__init__()
        assert inp_format in ("rel", "abs", "abs_rel")
        assert gan_type in ("probgan", "mgan", "infogan", "gan")
        if (inp_format != "rel" or gan_type not in ("mgan", "gan") or not global_disc
                or pool_type not in ("sways", "sgan") or num_discs != 1 or scene_dim <= 0):
            raise ValueError("HIP MultiDiscriminatorTrajectory implements the default hot path: inp_format='rel', "
                             "gan_type mgan/gan, global_disc, pool_type sways/sgan, one discriminator")
        self.pool_type = pool_type
        self.inp_format = inp_format
        self.unbound_output = unbound_output
        self.n_ds = num_discs
        self.gan_type = gan_type
        self.global_disc = global_disc
        self.inp_size = 2
        self.in_encoder = TrajectoryEncoder(hidden_size=h_dim, inp_size=self.inp_size, num_layers=1, embedding_dim=h_dim,
                                            return_hc=False)
        self.in_encoder_fc = nn.Sequential(nn.Linear(h_dim, h_dim // 2), nn.LeakyReLU(0.2),
                                           nn.Linear(h_dim // 2, h_dim // 2))
        self.pred_encoder = nn.Sequential(nn.Linear(pred_len * self.inp_size, h_dim), nn.LeakyReLU(0.2),
                                          nn.Linear(h_dim, h_dim // 2))
        if pool_type == "sways":
            self.social = SocialAttention(h_dim, h_dim)
        else:  # discriminators.py:62-67
            self.social = PoolHiddenNet(embedding_dim=16, h_dim=h_dim, mlp_dim=h_dim, bottleneck_dim=h_dim)
        h_dim *= 2
        self.scene_encoder = AttentionGlobal(noise_attention_dim=0, PhysFeature=True, num_layers=2, channels_cnn=8)
        h_dim += scene_dim
        self.discs = nn.ModuleList()
        for _ in range(num_discs):  # raw scores for the 'LS' / 'W' objectives (discriminators.py:76-85)
            layers = [nn.Linear(h_dim, h_dim // 2), nn.LeakyReLU(0.2), nn.Linear(h_dim // 2, 1)]
            self.discs.append(nn.Sequential(*(layers if unbound_output else layers + [nn.Sigmoid()])))
        if gan_type == "mgan":
            self.gen_id_reconstructor = nn.Sequential(nn.Linear(h_dim, h_dim // 2), nn.LeakyReLU(0.2),
                                                      nn.Linear(h_dim // 2, num_gens))
        self.eps = 1e-7
        self.len_hist = 1.0

    def _out_act(self):
        """Sigmoid + the eps squeeze of discriminators.py:83-84,203-204, or the raw score when unbound."""
        return HF.ACT_NONE if self.unbound_output else HF.ACT_SIGMOID_EPS

    def encode(self, in_xy, in_dxdy, pred_xy, pred_dxdy, mask=None):
        """(in_enc (b,h/2), pred_enc (K*b_m,h/2)) -> enc (K*b, h) as the reference returns it."""
        in_enc, pred_enc = self._encode_parts(in_dxdy, pred_dxdy)
        n_samples = pred_dxdy.shape[1]
        if mask is not None and not bool(mask.all()):
            pad = pred_enc.new_zeros(in_dxdy.size(1) * n_samples, pred_enc.size(1))
            pred_enc = pad.index_copy(0, mask.repeat(n_samples).nonzero().flatten(), pred_enc)
        return torch.cat([in_enc.repeat(n_samples, 1), pred_enc], dim=1)

    def history_context(self, in_dxdy, img, passes=1, lstm_branch=None, lstm_first=False, defer_cnn=None, scene_out=None):
        """(in_enc (b,h/2), scene (b,64)): everything that depends only on the observed history and the
        image crop.  The real and the fake pass of one discriminator step share it (identical inputs and
        weights), autograd sums their cotangents, so the history LSTM and the scene CNN run forward and
        backward once instead of twice; `passes` keeps the BatchNorm running-stat count of the reference.
        The scene CNN goes to branch stream 0; the history LSTM stays on the caller's stream or goes to branch
        stream `lstm_branch` (the caller joins both before it uses the results)."""
        self.ensure_flat()
        fc = self.in_encoder_fc
        def cnn():
            with HF.branch():  # scene CNN || history LSTM; joined by forward() right before the classifier input
                HF.mark("Dctx.cnn.begin")
                # (scene_out: an HF.OutSlot -- the attention head writes the features straight into the scene columns of the
                #  classifier input the caller has allocated, no broadcast launch behind the branch join)
                out = self.scene_encoder(img, stat_updates=passes, out=scene_out)
                if scene_out is not None:
                    out._mggan_X = scene_out.buf
                HF.mark("Dctx.cnn.end")
            return out

        def lstm():
            with HF.branch(lstm_branch) if lstm_branch is not None else contextlib.nullcontext():
                HF.mark("Dctx.lstm.begin")
                h = self.in_encoder(in_dxdy)
                out = HF.mlp(h, [(fc[0], HF.ACT_LEAKY, 0.2), (fc[2], HF.ACT_NONE, 0.0)])
                HF.mark("Dctx.lstm.end")
                if HF._on_branch():
                    # (a consumer that needs the history encoding only waits for THIS point of the branch stream, not for
                    #  the scene CNN queued behind it on the same stream: generator_step)
                    ev = torch.cuda.Event()
                    ev.record(HF._cur())
                    out._mggan_ready = ev
            return out

        if defer_cnn is not None:
            # the LSTM chain now; the scene CNN when the caller's hook fires (defer_cnn(callable): e.g. behind the generator's
            # sampling launches) -- the result arrives through the returned list
            in_enc = lstm()
            box = [in_enc, None]

            def late():
                with torch.no_grad():
                    box[1] = cnn()

            defer_cnn(late)
            return box
        if lstm_first:  # (both on one stream: the three small launches of the LSTM chain ahead of the CNN's persistent grids)
            in_enc = lstm()
            scene = cnn()
        else:
            scene = cnn()
            in_enc = lstm()
        return in_enc, scene

    def _encode_parts(self, in_dxdy, pred_dxdy, context=None):
        fc, pe = self.in_encoder_fc, self.pred_encoder
        if context is not None:
            in_enc = context[0]
        else:
            h = self.in_encoder(in_dxdy)
            in_enc = HF.mlp(h, [(fc[0], HF.ACT_LEAKY, 0.2), (fc[2], HF.ACT_NONE, 0.0)])
        _, n_samples, b, _ = pred_dxdy.shape
        x = pred_dxdy.permute(1, 2, 0, 3).reshape(n_samples * b, -1)
        pred_enc = HF.mlp(x, [(pe[0], HF.ACT_LEAKY, 0.2), (pe[2], HF.ACT_NONE, 0.0)])
        return in_enc, pred_enc

    def forward_pair(self, in_xy, in_dxdy, real_dxdy, fake_dxdy, seq_start_end, context):
        """The real and the fake single-sample pass of one discriminator step as ONE pass over 2b rows (rows
        [0,b) = real, [b,2b) = fake).  Every operator after the shared history context is row-wise or per
        scene, so the results equal two forward() calls; the latency-bound kernel chain runs once instead of twice.
        -> (out (2b,1): real rows then fake rows, branch_fake (b,1,g) | None)"""
        self.ensure_flat()
        in_enc, scene = context
        b = in_xy.size(1)
        cache = self.__dict__.setdefault("_pair_scenes", HF.BoundedCache(16))
        hit = cache.get(id(seq_start_end))
        fp = HF.scene_fingerprint(seq_start_end)
        if hit is None or hit[0] is not seq_start_end or (hit[2] != fp and len(hit) < 4):  # (4th item: a bucket's static lists)
            hit = cache.put(id(seq_start_end), (seq_start_end, [[int(s), int(e)] for s, e in seq_start_end] +
                                                [[int(s) + b, int(e) + b] for s, e in seq_start_end], fp))
        if self.pool_type == "sways":
            # one autograd node for the whole row pass (the row-pass nodes): rows [0,b) real, [b,2b) fake, both blocks with
            # social features (two independent single-sample passes), generator-id head on the fake half
            tb = HF.scene_tables(hit[1], 2 * b, in_enc.device)
            y, branch = self._row_pass(in_enc, scene, real_dxdy.reshape(real_dxdy.shape[0], b, 2),
                                       fake_dxdy.reshape(fake_dxdy.shape[0], b, 2), tb, 2, 2, b, in_xy[-1], in_dxdy[-1], b)
            return y, (branch.reshape(b, 1, -1) if branch is not None else None)
        pe = self.pred_encoder
        if real_dxdy.requires_grad or fake_dxdy.requires_grad:
            x = torch.cat([real_dxdy.reshape(real_dxdy.shape[0], b, 2).permute(1, 0, 2).reshape(b, -1),
                           fake_dxdy.reshape(fake_dxdy.shape[0], b, 2).permute(1, 0, 2).reshape(b, -1)], 0)
        else:  # (the discriminator step: both sets are constants) one launch instead of two permutes + cat
            x = HF.steps_to_rows(real_dxdy, fake_dxdy)
        pred_enc = HF.mlp(x, [(pe[0], HF.ACT_LEAKY, 0.2), (pe[2], HF.ACT_NONE, 0.0)])
        enc0 = torch.cat([in_enc.repeat(2, 1), pred_enc], dim=1)
        soc = self.social(in_xy[-1:], in_dxdy[-1:], enc0, hit[1], xy_mod=b)
        HF.join_branch(scene)
        classifier_inp = HF.DAssembleFn.apply(soc, in_enc, pred_enc, scene, 2, True)
        d = self.discs[0]
        head = [(d[0], HF.ACT_LEAKY, 0.2), (d[2], self._out_act(), 0.0)]
        if self.gan_type != "mgan":
            return HF.mlp(classifier_inp, head), None
        r = self.gen_id_reconstructor
        y, branch = HF.two_heads(classifier_inp, head, [(r[0], HF.ACT_LEAKY, 0.2), (r[2], HF.ACT_NONE, 0.0)], b)
        return y, branch.reshape(b, 1, -1)

    def _row_pass(self, in_enc, scene, pred, pred2, tb, K, soc_blocks, row0, xy_last, dxdy_last, xy_mod):
        """-> (score (K*b,1), id logits (K*b-row0, g) | None) through the two row-pass nodes (HF.DRowsBodyFn builds
        the classifier input in place, HF.DRowsHeadsFn adds the scene block and runs the heads)."""
        anchor = self.discs[0][0].weight
        save = HF.want_grad(in_enc, scene, pred, pred2, anchor)
        if pred2 is None and HF.d_rows_lean_ok(self, in_enc, scene, pred, K, soc_blocks, row0):
            # frozen discriminator, K >= 2 sample blocks: block 0 through the generic kernels, the others in one launch
            return HF.DRowsLeanFn.apply(in_enc, scene, pred, self, tb, K, xy_last, dxdy_last, save)
        lean = HF.dheads_lean_ok(self, in_enc, scene, K, soc_blocks, row0)
        X = HF.DRowsBodyFn.apply(in_enc, pred, pred2, anchor, self, tb, K, soc_blocks, xy_last, dxdy_last, xy_mod,
                                 scene.shape[1], save, lean)
        HF.join_branch(scene)  # the scene CNN's branch only has to be there now
        return HF.DRowsHeadsFn.apply(X, scene, anchor, self, K, row0, save, lean)

    def forward(self, in_xy, in_dxdy, pred_xy, pred_dxdy, seq_start_end, return_all=False, img=None, mask=None,
                context=None):
        """Returns output (b_m, K) and, for gan_type 'mgan', branch_out (b_m, K, num_gens)."""
        if img is None:
            raise ValueError("img is mandatory: scene_dim=64 is hard-wired into the model (SURVEY A.6)")
        self.ensure_flat()
        if pred_xy.dim() == 3:
            pred_xy, pred_dxdy = pred_xy.unsqueeze(1), pred_dxdy.unsqueeze(1)
        pred_len, n_samples, b, _ = pred_xy.shape
        full_b = in_xy.size(1)
        masked = HF.is_masked(mask)  # pass mask=None (all valid) to avoid the sync

        if context is not None and masked:
            raise ValueError("a shared history context needs mask=None (all pedestrians valid)")
        if not masked and self.pool_type == "sways" and n_samples * b > 0:
            # the fused row pass: social features for sample block 0 only -- `seq_start_end * n_samples` is LIST
            # repetition in the reference
            if context is not None:
                in_enc, scene = context
            else:
                fc = self.in_encoder_fc
                in_enc = HF.mlp(self.in_encoder(in_dxdy), [(fc[0], HF.ACT_LEAKY, 0.2), (fc[2], HF.ACT_NONE, 0.0)])
                scene = self.scene_encoder(img)
            tb = HF.scene_tables(seq_start_end, full_b, in_enc.device)
            y, branch_out = self._row_pass(in_enc, scene, pred_dxdy, None, tb, n_samples, 1, 0, in_xy[-1], in_dxdy[-1], 0)
            output = y.reshape(n_samples, b).t()  # mean over the single discriminator is the identity
            if self.gan_type == "gan":
                return output
            return output, branch_out.reshape(n_samples, b, -1).transpose(0, 1)
        in_enc, pred_enc = self._encode_parts(in_dxdy, pred_dxdy, context)
        if not masked:
            # social features only for sample block 0: `seq_start_end * n_samples` is LIST repetition (A.1)
            enc0 = torch.cat([in_enc, pred_enc if n_samples == 1 else pred_enc[:full_b]], dim=1)
            soc0 = self.social(in_xy, in_dxdy, enc0, seq_start_end)
            scene = context[1] if context is not None else self.scene_encoder(img)
            HF.join_branch(scene)
            # sways: only sample block 0 carries social features (A.1); sgan: PoolHiddenNet walks the repeated list
            # and returns K copies of the block-0 result (mode 2 broadcasts it)
            classifier_inp = HF.DAssembleFn.apply(soc0, in_enc, pred_enc, scene, n_samples,
                                                  2 if self.pool_type == "sgan" else False)
        else:
            enc = self.encode(in_xy, in_dxdy, pred_xy, pred_dxdy, mask)
            soc = self.social(in_xy.repeat(1, n_samples, 1), in_dxdy.repeat(1, n_samples, 1), enc,
                              seq_start_end * n_samples)
            classifier_inp = torch.cat([soc, enc], dim=1)[mask.repeat(n_samples)]
            scene = self.scene_encoder(img[mask]).repeat(n_samples, 1)
            classifier_inp = torch.cat([classifier_inp, scene], 1)

        d = self.discs[0]
        head = [(d[0], HF.ACT_LEAKY, 0.2), (d[2], self._out_act(), 0.0)]
        if self.gan_type == "gan":
            return HF.mlp(classifier_inp, head).reshape(n_samples, b).t()
        r = self.gen_id_reconstructor
        y, branch_out = HF.two_heads(classifier_inp, head, [(r[0], HF.ACT_LEAKY, 0.2), (r[2], HF.ACT_NONE, 0.0)])
        output = y.reshape(n_samples, b).t()  # mean over the single discriminator is the identity
        return output, branch_out.reshape(n_samples, b, -1).transpose(0, 1)
