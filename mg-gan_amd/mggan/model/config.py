"""Command-line surface of the reference (/root/reference/mggan/model/config.py:4-133), on plain
argparse (test_tube's HyperOptArgumentParser is only used as an ArgumentParser there).
Additions: dataset choice 'synthetic' (+ its shape flags) and --rng."""
import argparse


BUILT_WIDTHS = {"h_dim": 32, "decoder_h_dim": 32}


def check_widths(config):
    """The kernels of libmggan_hip.so are instantiated for the reference's DEFAULT widths (config.py:70-71 there:
    --h_dim 32 -> generator encoder / social features 32, discriminator encoder 64; --decoder_h_dim 32 -> rollout LSTM 32,
    step embedding 16) and for noise vectors whose length is a multiple of 4.  NARROWER models (2 <= width <= 32) run on the
    same kernels zero-padded (mggan/model/widths.py); wider ones are
    refused HERE -- at parse time and again in construct_model -- with a ValueError, before a module is built or a kernel
    launched."""
    bad = []
    narrow = False
    for flag, built in BUILT_WIDTHS.items():
        v = int(getattr(config, flag, built))
        if not 2 <= v <= built:
            bad.append("--{} {} (built: {}, narrower widths run zero-padded)".format(flag, v, built))
        narrow |= v != built
    z = int(getattr(config, "noise_dim", 8))
    if z < 4 or z % 4:
        bad.append("--noise_dim {} (built: positive multiples of 4)".format(z))
    if narrow and not bad and int(getattr(config, "n_social_modules", 1)) <= 0:
        bad.append("--n_social_modules 0 at a narrower width")
    if bad:
        raise ValueError("not built on the HIP path: " + "; ".join(bad) + ".  libmggan_hip.so instantiates its LSTM, social-"
                         "attention and rollout kernels for the reference's default widths (DESIGN.md section 9).")
    return config


class _Parser(argparse.ArgumentParser):
    def opt_list(self, *args, options=None, tunable=None, **kwargs):
        return self.add_argument(*args, **kwargs)

    def parse_known_args(self, args=None, namespace=None):  # (parse_args goes through here too)
        ns, rest = super().parse_known_args(args, namespace)
        return check_widths(ns), rest


def get_parser():
    parser = _Parser()
    parser.add_argument("--name", type=str, default="test")
    parser.add_argument("--log_dir", type=str, default="./logs/")
    # the reference's default ("stanford_synthetic") and "social_stanford_synthetic" need the occupancy-map variants of
    # SDD, which are out of scope: they are rejected here, at parse time, and the default is the
    # built-in synthetic dataset so that `train.py --name test` runs as in the reference's README
    parser.add_argument("--dataset", type=str, default="synthetic",
                        choices=["hotel", "eth", "zara1", "zara2", "univ", "stanford", "gofp", "synthetic"])
    parser.add_argument("--gpus", type=str, default="0")
    parser.add_argument("--workers", type=int, default=0)
    parser.add_argument("--batch_size", type=int, default=2)
    parser.opt_list("--beta1", type=float, default=0.5, options=[0.1, 0.5, 0.9], tunable=False)
    parser.add_argument("--l2_loss_weight", type=float, default=1.0)
    parser.add_argument("--clf_loss_weight", type=float, default=1.0)
    parser.add_argument("--pi_net_loss_weight", type=float, default=1.0)
    parser.add_argument("--epochs", type=int, default=500)
    parser.add_argument("--clipping_threshold_d", type=int, default=100)
    parser.add_argument("--clipping_threshold_g", type=int, default=500)
    parser.add_argument("--num_gen_steps", type=int, default=1)
    parser.add_argument("--inp_format", choices=["rel", "abs", "abs_rel"], default="rel")
    parser.add_argument("--keep_gen_steps", type=int, default=0)
    parser.add_argument("--top_k_test", type=int, default=20)
    parser.add_argument("--val_every", type=int, default=1)
    parser.add_argument("--save_every", type=int, default=5)
    parser.add_argument("--num_unrolling_steps", type=int, default=0)
    parser.add_argument("--debug", action="store_true")
    parser.add_argument("--n_social_modules", type=int, default=1)
    parser.add_argument("--g_lr", type=float, default=1e-3)
    parser.add_argument("--d_lr", type=float, default=1e-3)
    parser.add_argument("--sigma", type=float, default=1.0)
    parser.add_argument("--gan_type", type=str, choices=["probgan", "mgan", "infogan", "gan"], default="mgan")
    parser.add_argument("--experiment", type=str, choices=["multi_generator", "discrete"], default="multi_generator")
    parser.add_argument("--pool_type", type=str, default="sways")
    parser.add_argument("--global_disc", type=int, default=1)
    parser.add_argument("--unconditional", action="store_true")
    parser.add_argument("--augment", type=int, default=1)
    parser.add_argument("--noise_dim", type=int, default=8)
    parser.add_argument("--h_dim", type=int, default=32)
    parser.add_argument("--decoder_h_dim", type=int, default=32)
    parser.add_argument("--num_samples", type=int, default=20)
    parser.add_argument("--num_expectation_samples", type=int, default=1)
    parser.add_argument("--weighting_target", type=str, choices=["l2", "disc_scores", "endpoint", "mgan", "ml", "none"],
                        default="ml")
    parser.opt_list("--l2_loss_type", type=str, options=["none", "min_z", "min_g_z", "min_g_min_z", "mse"],
                    default="min_g_z", tunable=False)
    parser.opt_list("--num_gens", type=int, options=[2, 3, 4, 5], default=1, tunable=True)
    parser.opt_list("--l2_decay_rate", type=float, options=[1, 0.99, 0.9], default=1, tunable=False)
    parser.add_argument("--checkpoint", type=str)
    parser.opt_list("--sghmc_alpha", default=0.01, type=float, dest="sghmc_alpha", options=[0.1, 0.01, 0.001],
                    tunable=False)
    parser.add_argument("--g_noise_loss_lambda", default=3e-2, type=float, dest="g_noise_loss_lambda")
    parser.add_argument("--d_noise_loss_lambda", default=3e-2, type=float, dest="d_noise_loss_lambda")
    parser.add_argument("--d_hist_loss_lambda", default=1.0, type=float, dest="d_hist_loss_lambda")
    parser.opt_list("--gan_obj", default="NS", type=str, dest="gan_obj", options=["NS", "MM", "LS", "W"], tunable=False)
    # ---- additions of the MI355X build (not in the reference CLI) ----
    parser.add_argument("--rng", type=str, choices=["host", "device"], default="device",
                        help="device (default): every random number of an iteration from one Philox launch, no host sync "
                             "-- train() replays captured iterations; host: the reference's draw order on the CPU "
                             "generators (seed-comparable with the reference; eager launches, one read-back per generator call)")
    parser.add_argument("--bn_sync", type=str, choices=["global", "local"], default="global",
                        help="sharded runs: BatchNorm statistics over the global batch (all-reduced: same results as one "
                             "process) or per rank (what DistributedDataParallel does without SyncBatchNorm)")
    parser.add_argument("--graph", type=str, choices=["auto", "on", "off"], default="auto",
                        help="train(): replay the iteration as a captured HIP graph per batch shape (needs --rng device; "
                             "auto = on whenever the configuration allows it), eager launches otherwise")
    parser.add_argument("--graph_shapes", type=int, default=8, help="batch shapes the graph cache of train() keeps")
    parser.add_argument("--graph_buckets", type=int, default=64, help="shape buckets of padded (ragged) batches the cache keeps")
    parser.add_argument("--graph_pad", type=str, choices=["auto", "on", "off"], default="auto",
                        help="train(): pad RAGGED batches (the reference loader's: a new tuple of scene sizes almost every "
                             "batch) to shape buckets with inert phantom pedestrians, so that one captured graph per bucket "
                             "replays them all.  auto: whenever graphs are replayed; on: also with --graph off (eager "
                             "launches on the padded batches: what a replay is bit-identical to); off: exact shapes only")
    parser.add_argument("--graph_bucket", type=str, choices=["quarter", "pow2"], default="quarter",
                        help="bucket sizes of --graph_pad: four per octave (at most a quarter of a bucket is padding) or "
                             "powers of two (fewer graphs, up to half)")
    parser.add_argument("--crop_device", type=str, choices=["auto", "on", "off"], default="auto",
                        help="on-disk datasets: cut the per-pedestrian scene crops on the GPU from scene images resident in HBM "
                             "(training augmentation included, bit-identical to the host's Pillow path) instead of on the "
                             "host.  auto = on")
    parser.add_argument("--cache_device", type=int, default=0,
                        help="synthetic dataset: keep the produced batches resident in HBM after their first use")
    parser.add_argument("--synthetic_scenes", type=int, default=64, help="scenes per synthetic epoch")
    parser.add_argument("--synthetic_peds", type=int, default=0, help="pedestrians per scene (0 = ragged 1..6)")
    return parser
