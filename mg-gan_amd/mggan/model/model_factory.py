"""construct_model(config) -> (G, D)   (surface of /root/reference/mggan/model/model_factory.py:7-86)."""
from mggan.utils import count_parameters
from mggan.model.modules.standard import MultiGenerator
from mggan.model.modules.discriminators import MultiDiscriminatorTrajectory


def construct_model(config):
    from mggan.model.config import check_widths

    check_widths(config)  # ValueError for widths the kernels are not instantiated for (a hand-built namespace lands here)
    unbound_output = config.gan_obj in ["W", "LS"]
    num_discs = 5 if config.gan_type == "probgan" else 1
    config.use_pinet = config.weighting_target != "none" and not config.unconditional
    pred_len = 12
    scene_dim = 8 * 8
    if config.experiment not in ("multi_generator", "discrete"):
        raise ValueError("Requested model not implemented on the HIP path ('multi_generator' and 'discrete' are).")
    common = dict(z_size=config.noise_dim, inp_format=config.inp_format, encoder_h_dim=config.h_dim,
                  decoder_h_dim=config.decoder_h_dim, social_feat_size=config.h_dim if config.n_social_modules > 0 else 0,
                  num_gens=config.num_gens, pred_len=pred_len, pool_type=config.pool_type,
                  num_social_modules=config.n_social_modules, scene_dim=scene_dim, use_pinet=config.use_pinet,
                  learn_prior=config.unconditional)
    if config.experiment == "discrete":  # model_factory.py:50-66
        from mggan.model.modules.standard_discrete import DiscreteLatentGenerator

        G = DiscreteLatentGenerator(embedding_dim=16, **common)
    else:
        G = MultiGenerator(embedding_dim=int(config.decoder_h_dim // 2), **common)
    D = MultiDiscriminatorTrajectory(num_discs=num_discs, num_gens=config.num_gens, unbound_output=unbound_output,
                                     h_dim=config.h_dim * 2, pred_len=pred_len, inp_format=config.inp_format,
                                     gan_type=config.gan_type, scene_dim=scene_dim, global_disc=config.global_disc,
                                     pool_type=config.pool_type)
    print("G #parameters: ", count_parameters(G))
    print("D #parameters: ", count_parameters(D))
    config.num_gen_parameters = count_parameters(G)
    return G, D
