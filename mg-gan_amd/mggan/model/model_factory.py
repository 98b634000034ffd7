"""construct_model(config) -> (G, D)   (surface of /root/reference/mggan/model/model_factory.py:7-86)."""
from mggan.utils import count_parameters
from mggan.model.modules.standard import MultiGenerator
from mggan.model.modules.discriminators import MultiDiscriminatorTrajectory


def _build(config, h_dim, decoder_h_dim, num_discs, unbound_output, pred_len=12, scene_dim=8 * 8):
    common = dict(z_size=config.noise_dim, inp_format=config.inp_format, encoder_h_dim=h_dim,
                  decoder_h_dim=decoder_h_dim, social_feat_size=h_dim if config.n_social_modules > 0 else 0,
                  num_gens=config.num_gens, pred_len=pred_len, pool_type=config.pool_type,
                  num_social_modules=config.n_social_modules, scene_dim=scene_dim, use_pinet=config.use_pinet,
                  learn_prior=config.unconditional)
    if config.experiment == "discrete":  # model_factory.py:50-66
        from mggan.model.modules.standard_discrete import DiscreteLatentGenerator

        G = DiscreteLatentGenerator(embedding_dim=16, **common)
    else:
        G = MultiGenerator(embedding_dim=int(decoder_h_dim // 2), **common)
    D = MultiDiscriminatorTrajectory(num_discs=num_discs, num_gens=config.num_gens, unbound_output=unbound_output,
                                     h_dim=h_dim * 2, pred_len=pred_len, inp_format=config.inp_format,
                                     gan_type=config.gan_type, scene_dim=scene_dim, global_disc=config.global_disc,
                                     pool_type=config.pool_type)
    return G, D


def construct_model(config):
    import torch

    from mggan.model import widths
    from mggan.model.config import check_widths

    check_widths(config)  # ValueError for widths the kernels are not instantiated for (a hand-built namespace lands here)
    unbound_output = config.gan_obj in ["W", "LS"]
    num_discs = 5 if config.gan_type == "probgan" else 1
    config.use_pinet = config.weighting_target != "none" and not config.unconditional
    if config.experiment not in ("multi_generator", "discrete"):
        raise ValueError("Requested model not implemented on the HIP path ('multi_generator' and 'discrete' are).")
    h, dh = int(config.h_dim), int(config.decoder_h_dim)
    if (h, dh) == (widths.BUILT_H, widths.BUILT_DH):
        G, D = _build(config, h, dh, num_discs, unbound_output)
    else:
        # narrower model: the seeded initial values are drawn at the reference's shapes, in its order (so a seed gives the
        # reference's model), then scattered into modules built at the kernels' widths; mggan/model/widths.py
        with widths.logical_holders():
            Gl, Dl = _build(config, h, dh, num_discs, unbound_output)
        rng_state = torch.get_rng_state()
        G, D = _build(config, widths.BUILT_H, widths.BUILT_DH, num_discs, unbound_output)
        torch.set_rng_state(rng_state)
        widths.attach(G, widths.generator_rules(h, dh, int(config.noise_dim), config.pool_type, config.experiment == "discrete"))
        widths.attach(D, widths.discriminator_rules(h, config.pool_type))
        G.load_state_dict(Gl.state_dict())
        D.load_state_dict(Dl.state_dict())
        assert widths.padding_is_zero(G) and widths.padding_is_zero(D)
    print("G #parameters: ", widths.logical_parameter_count(G))
    print("D #parameters: ", widths.logical_parameter_count(D))
    config.num_gen_parameters = widths.logical_parameter_count(G)
    return G, D
