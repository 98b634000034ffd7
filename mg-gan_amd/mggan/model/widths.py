"""Model widths below the built ones (--h_dim / --decoder_h_dim < 32, config.py:70-71 of the reference).

libmggan_hip.so instantiates its LSTM, social-attention and rollout kernels for ONE set of widths (generator encoder /
social features 32, discriminator encoder 64, rollout LSTM 32, step embedding 16 ...).  A narrower model runs on those
kernels ZERO-PADDED: the modules are built at the physical widths, every parameter whose shape depends on --h_dim /
--decoder_h_dim holds the logical tensor in a fixed set of rows / columns and exact zeros elsewhere.  A padded unit has a
zero pre-activation (LSTM: i = f = o = 1/2, g = 0 -> c = h = 0; ReLU / LeakyReLU: 0), zero outgoing weights and therefore a
zero cotangent, so its gradients are exactly zero, AdamW (decay included) keeps it at zero, and the gradient norm the
clipping sees is the logical model's: the padded model IS the narrow model, computed on wider tiles.

The reference's shapes show at the checkpoint surface only: state_dict() / load_state_dict() (and the optimizers'
moments) gather / scatter through the block tables below, so checkpoints are interchangeable with the reference's at the
same flags.  Layouts follow the reference's concatenations: standard.py:142-155 (enc_h = [lstm | scene | social]),
:247 ([enc_h | noise]), common_modules.py:122 ([h_dec | social]), discriminators.py:141,185,196 ([soc | in | pred | scene])."""
import contextlib
import re

import torch

BUILT_H, BUILT_DH = 32, 32
_HOLDERS = [False]


def holders_only():
    """True while construct_model builds the LOGICAL-shape modules it only takes the seeded initial values from (the
    constructors' width checks are skipped; such a module never launches a kernel)."""
    return _HOLDERS[0]


@contextlib.contextmanager
def logical_holders():
    _HOLDERS[0] = True
    try:
        yield
    finally:
        _HOLDERS[0] = False


def _p(pw, lw):
    return [(pw, lw)]


def _gates(pw, lw):
    return [(pw, lw)] * 4


def generator_rules(h, dh, z, pool_type="sways", discrete=False):
    """MultiGenerator (standard.py:17-109) or DiscreteLatentGenerator (standard_discrete.py:18-106: ONE decoder, embeddings of
    16 whatever --decoder_h_dim says, the encoded generator id between enc_h and the noise)."""
    e = 16 if discrete else dh // 2
    hm, dm = h // 2, dh // 2
    dec = r"decoder\." if discrete else r"(?:gs\.\d+|G_\d+)\."
    if pool_type == "sways":
        social = [
            (r"social\.feature_embedder\.fc\.4\.(weight|bias)", _p(32, h), None),
            (r"social\.attention\.W\.weight", _p(32, h), _p(32, h)),
            (r"social\.attention\.W\.bias", _p(32, h), None),
        ]
    else:  # PoolHiddenNet (social_gan.py:157-229): [position embedding | hidden] -> h -> bottleneck = h, max over the scene
        social = [
            (r"social\.spatial_embedding\.(weight|bias)", _p(16, e), None),
            (r"social\.mlp_pre_pool\.0\.weight", _p(32, h), [(16, e), (32, h)]),
            (r"social\.mlp_pre_pool\.0\.bias", _p(32, h), None),
            (r"social\.mlp_pre_pool\.2\.weight", _p(32, h), _p(32, h)),
            (r"social\.mlp_pre_pool\.2\.bias", _p(32, h), None),
        ]
    return social + [
        (r"encoder\.embedding\.(weight|bias)", _p(16, e), None),
        (r"encoder\.encoder\.weight_ih_l0", _gates(32, h), _p(16, e)),
        (r"encoder\.encoder\.weight_hh_l0", _gates(32, h), _p(32, h)),
        (r"encoder\.encoder\.bias_[ih]h_l0", _gates(32, h), None),
        (dec + r"decoder\.weight_ih_l0", _gates(32, dh), _p(16, e)),
        (dec + r"decoder\.weight_hh_l0", _gates(32, dh), _p(32, dh)),
        (dec + r"decoder\.bias_[ih]h_l0", _gates(32, dh), None),
        (dec + r"spatial_embedding\.(weight|bias)", _p(16, e), None),
        (dec + r"hidden2pos\.0\.weight", _p(16, dm), [(32, dh), (32, h)]),
        (dec + r"hidden2pos\.0\.bias", _p(16, dm), None),
        (dec + r"hidden2pos\.2\.weight", None, _p(16, dm)),
        (r"enc_h_to_dec_h\.0\.weight", _p(32, dh), [(32, h), (64, 64), (32, h), (z, z)] + ([(z, z)] if discrete else [])),
        (r"enc_h_to_dec_h\.0\.bias", _p(32, dh), None),
        (r"net_chooser\.0\.weight", _p(16, hm), [(32, h), (64, 64), (32, h)]),
        (r"net_chooser\.0\.bias", _p(16, hm), None),
        (r"net_chooser\.2\.weight", _p(16, hm), _p(16, hm)),
        (r"net_chooser\.2\.bias", _p(16, hm), None),
        (r"net_chooser\.4\.weight", None, _p(16, hm)),
    ]


def discriminator_rules(h, pool_type="sways"):
    H, Q = 2 * h, h           # discriminators.py: h_dim = 2 * --h_dim; h_dim // 2
    M = (2 * H + 64) // 2     # hidden width of the classifier heads (:77-80)
    enc = [(32, Q), (32, Q)]  # [in | pred]: the layout of `enc` and of everything that is a weighted sum of its rows
    head = r"(?:discs\.\d+|gen_id_reconstructor)\."
    if pool_type == "sways":
        soc = enc             # attention pooling: soc = attention @ enc
        social = [
            (r"social\.feature_embedder\.fc\.4\.(weight|bias)", _p(64, H), None),
            (r"social\.attention\.W\.weight", _p(64, H), enc),
            (r"social\.attention\.W\.bias", _p(64, H), None),
        ]
    else:                     # PoolHiddenNet(embedding_dim=16, h_dim=H, bottleneck_dim=H) (discriminators.py:62-67)
        soc = _p(64, H)
        social = [
            (r"social\.mlp_pre_pool\.0\.weight", _p(64, H), [(16, 16)] + enc),
            (r"social\.mlp_pre_pool\.0\.bias", _p(64, H), None),
            (r"social\.mlp_pre_pool\.2\.weight", _p(64, H), _p(64, H)),
            (r"social\.mlp_pre_pool\.2\.bias", _p(64, H), None),
        ]
    return social + [
        (r"in_encoder\.embedding\.(weight|bias)", _p(64, H), None),
        (r"in_encoder\.encoder\.weight_[ih]h_l0", _gates(64, H), _p(64, H)),
        (r"in_encoder\.encoder\.bias_[ih]h_l0", _gates(64, H), None),
        (r"in_encoder_fc\.0\.weight", _p(32, Q), _p(64, H)),
        (r"in_encoder_fc\.0\.bias", _p(32, Q), None),
        (r"in_encoder_fc\.2\.weight", _p(32, Q), _p(32, Q)),
        (r"in_encoder_fc\.2\.bias", _p(32, Q), None),
        (r"pred_encoder\.0\.(weight|bias)", _p(64, H), None),
        (r"pred_encoder\.2\.weight", _p(32, Q), _p(64, H)),
        (r"pred_encoder\.2\.bias", _p(32, Q), None),
        (head + r"0\.weight", _p(96, M), soc + enc + [(64, 64)]),  # [soc | in | pred | scene]
        (head + r"0\.bias", _p(96, M), None),
        (head + r"2\.weight", None, _p(96, M)),
    ]


def _index(blocks):
    idx, off = [], 0
    for pw, lw in blocks:
        assert 0 < lw <= pw, (pw, lw)
        idx.append(torch.arange(off, off + lw))
        off += pw
    return torch.cat(idx), off


class WidthMap:
    """name -> (row index, column index) of the logical tensor inside the physical parameter."""

    def __init__(self, rules):
        self.rules = [(re.compile(pat + r"$"), r, c) for pat, r, c in rules]
        self._cache = {}

    def lookup(self, name, phys_shape):
        hit = self._cache.get(name)
        if hit is None:
            hit = (None, None)
            for pat, rows, cols in self.rules:
                if pat.match(name):
                    ri = ci = None
                    if rows is not None:
                        ri, n = _index(rows)
                        assert n == phys_shape[0], (name, n, tuple(phys_shape))
                    if cols is not None:
                        ci, n = _index(cols)
                        assert n == phys_shape[1], (name, n, tuple(phys_shape))
                    hit = (ri, ci)
                    break
            self._cache[name] = hit
        return hit

    def logical_shape(self, name, phys_shape):
        ri, ci = self.lookup(name, phys_shape)
        s = list(phys_shape)
        if ri is not None:
            s[0] = ri.numel()
        if ci is not None:
            s[1] = ci.numel()
        return tuple(s)

    def to_logical(self, name, t):
        ri, ci = self.lookup(name, t.shape)
        if ri is not None:
            t = t.index_select(0, ri.to(t.device))
        if ci is not None:
            t = t.index_select(1, ci.to(t.device))
        return t

    def to_physical(self, name, t, phys_shape):
        ri, ci = self.lookup(name, phys_shape)
        if ri is None and ci is None:
            return t
        if tuple(t.shape) == tuple(phys_shape):  # already physical (a state_dict taken with the hooks off)
            return t
        want = self.logical_shape(name, phys_shape)
        if tuple(t.shape) != want:
            raise RuntimeError("size mismatch for {}: checkpoint {} vs the model's logical shape {} (physical {})".format(
                name, tuple(t.shape), want, tuple(phys_shape)))
        out = torch.zeros(phys_shape, dtype=t.dtype, device=t.device)
        if ri is not None and ci is not None:
            out[ri.to(t.device)[:, None], ci.to(t.device)[None, :]] = t
        elif ri is not None:
            out[ri.to(t.device)] = t
        else:
            out[:, ci.to(t.device)] = t
        return out

    def padding_mask(self, name, phys_shape):
        """bool tensor, True where the physical parameter is padding (must stay exactly zero)."""
        ri, ci = self.lookup(name, phys_shape)
        m = torch.ones(phys_shape, dtype=torch.bool)
        if ri is None and ci is None:
            return ~m
        rows = ri if ri is not None else torch.arange(phys_shape[0])
        if len(phys_shape) == 1:
            m[rows] = False
        else:
            cols = ci if ci is not None else torch.arange(phys_shape[1])
            m[rows[:, None], cols[None, :]] = False
        return m


def attach(module, rules):
    """Give `module` (a root: the generator or the discriminator) the reference's shapes at state_dict() /
    load_state_dict(); its parameters stay physical."""
    wm = WidthMap(rules)
    module._width_map = wm

    def save_hook(mod, sd, prefix, local_metadata):
        shapes = {prefix + k: v.shape for k, v in mod.named_parameters(remove_duplicate=False)}
        for k in list(sd.keys()):
            if k in shapes:
                sd[k] = wm.to_logical(k[len(prefix):], sd[k])
        return sd

    def load_hook(sd, prefix, local_metadata, strict, missing, unexpected, errors):
        for k, p in module.named_parameters(remove_duplicate=False):
            if prefix + k in sd:
                try:
                    sd[prefix + k] = wm.to_physical(k, sd[prefix + k], p.shape)
                except RuntimeError as e:
                    errors.append(str(e))
                    del sd[prefix + k]

    module._register_state_dict_hook(save_hook)
    module._register_load_state_dict_pre_hook(load_hook)
    return wm


def logical_parameter_count(module):
    wm = getattr(module, "_width_map", None)
    n = 0
    for k, p in module.named_parameters():
        if not p.requires_grad:
            continue
        shape = wm.logical_shape(k, p.shape) if wm is not None else p.shape
        c = 1
        for s in shape:
            c *= s
        n += c
    return n


def padding_is_zero(module):
    """The invariant the whole scheme rests on (tests assert it after training iterations)."""
    wm = getattr(module, "_width_map", None)
    if wm is None:
        return True
    for k, p in module.named_parameters():
        m = wm.padding_mask(k, p.shape).to(p.device)
        if bool(m.any()) and bool((p.detach()[m] != 0).any()):
            return False
    return True
