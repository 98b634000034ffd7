"""Golden vectors for MASKED batches (pedestrians whose ground truth contains NaN, GOFP only in the reference:
trajectories_scene.py:169-174): one D+G+PM iteration of the REAL reference on CPU with the loss mask of
abstract_train.py:127-132, recording the draws, the logged losses and the parameters after the iteration.
Pins SURVEY A.10 (the L2 term slices the masked predictions with the unmasked scene bounds and divides by the
unmasked b).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_masked.py [masked] [variants] [sgan]
        -> tests/golden/golden_masked_g2.npz, golden_{obj_ls,obj_mm,wt_l2,wt_endpoint,wt_mgan,pool_sgan}_g2.npz
(`variants`: one unmasked iteration each under --gan_obj LS / MM and --weighting_target l2 / endpoint)."""
import importlib.util
import os
import sys
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

ref_train, ref_config = _refload.load_reference()
import test_tube  # noqa: E402  (stub)
import mggan.model.modules.standard as ref_standard  # noqa: E402
import mggan.model.modules.standard_discrete as ref_discrete  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "_synth", os.path.join(HERE, "..", "..", "mg-gan_amd", "mggan", "data_utils", "synthetic.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def main(tag="masked_g2", extra=(), nan=True, num_gens=2, sizes=(2, 3, 1, 4), seed=21, keep_init=True):
    out = {}
    args = ref_config.get_parser().parse_args(["--gpus", "", "--num_gens", str(num_gens)] + list(extra))
    torch.manual_seed(seed)
    np.random.seed(seed + 1)
    G, D = ref_train.construct_model(args)
    model = ref_train.PiNetMultiGeneratorGAN(G, D, args, test_tube.Experiment())
    G.train()
    D.train()
    if keep_init:  # (the variants re-create the initial state from the seed: seeded init is bit-identical)
        for pre, mod in (("G0", G), ("D0", D)):
            for k, v in mod.state_dict().items():
                out["{}/{}".format(pre, k)] = t2n(v)
    batch = synth.make_batch(list(sizes), seed=seed + 2)
    if nan:
        batch["gt_xy"][5:, 3] = float("nan")    # pedestrian 3 (second scene) leaves after 5 predicted steps
        batch["gt_dxdy"][5:, 3] = float("nan")
        batch["gt_xy"][9:, 8] = float("nan")    # pedestrian 8 (last scene)
        batch["gt_dxdy"][9:, 8] = float("nan")
    sub = batch["seq_start_end"]
    in_xy, in_dxdy, gt_xy, gt_dxdy, img = (batch[k] for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy", "features"))
    mask = ~gt_xy.isnan().any(2).any(0)     # abstract_train.py:127
    out["meta/num_gens"], out["meta/scenes"] = np.int64(num_gens), np.array(sub, dtype=np.int64)
    out["meta/seed"], out["meta/args"] = np.int64(seed), np.array(list(extra))
    for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy", "features"):
        out["in/" + k] = t2n(batch[k])
    out["in/mask"] = t2n(mask)

    rec = {}
    orig_labels = ref_train.get_gan_labels

    def labels(shape, smoothness=0.1):
        lr, lf = orig_labels(shape, smoothness)
        rec.setdefault("labels", []).append((float(lr.flatten()[0]), float(lf.flatten()[0])))
        return lr, lf

    ref_train.get_gan_labels = labels
    orig_noise = ref_standard.get_global_noise

    def noise_std(dim, sb, kind):
        n = orig_noise(dim, sb, kind)
        rec.setdefault("inner_noise", []).append(n)
        return n

    ref_standard.get_global_noise = noise_std
    ref_discrete.get_global_noise = noise_std
    orig_fwd = G.forward

    def fwd(*a, **kw):
        o = orig_fwd(*a, **kw)
        rec["G_noise"], rec["G_idx"] = kw.get("noise"), o[2]
        return o

    G.forward = fwd
    step_args = (in_xy, in_dxdy, gt_xy[:, mask], gt_dxdy[:, mask], sub)   # abstract_train.py:128-131
    for si, step in enumerate(("d", "g", "pm")):
        rec.clear()
        torch.manual_seed(seed + 100 + si)
        np.random.seed(seed + 150 + si)
        metrics = defaultdict(list)
        getattr(model, {"d": "discriminator_step", "g": "generator_step", "pm": "net_chooser_step"}[step])(
            *step_args, metrics, mask, img)
        p = "s_" + step
        for k, v in metrics.items():
            out[p + "/metric/" + k] = np.float64(v[0])
        if "labels" in rec:
            out[p + "/labels"] = np.array(rec["labels"], dtype=np.float64)
        out[p + "/noise"] = t2n(torch.stack(rec["inner_noise"]) if step == "pm" else rec["G_noise"])
        out[p + "/gen_idxs"] = t2n(rec["G_idx"])
        print(step, {k: v[0] for k, v in metrics.items() if "probs" not in k})
    for pre, mod in (("G1", G), ("D1", D)):
        for k, v in mod.state_dict().items():
            out["{}/{}".format(pre, k)] = t2n(v)
    ref_train.get_gan_labels = orig_labels
    ref_standard.get_global_noise = orig_noise
    ref_discrete.get_global_noise = orig_noise
    path = os.path.join(HERE, "golden_{}.npz".format(tag))
    np.savez_compressed(path, **out)
    print(tag, "bytes", os.path.getsize(path))


if __name__ == "__main__":
    torch.set_num_threads(1)
    which = sys.argv[1:] or ["masked", "variants"]
    if "masked" in which:
        main()
    if "variants" in which:  # other GAN objectives and PM-network targets, one unmasked iteration each
        for tag, extra in (("obj_ls_g2", ["--gan_obj", "LS"]), ("obj_mm_g2", ["--gan_obj", "MM"]),
                           ("wt_l2_g2", ["--weighting_target", "l2"]), ("wt_endpoint_g2", ["--weighting_target", "endpoint"]),
                           ("wt_mgan_g2", ["--weighting_target", "mgan"])):
            main(tag, extra, nan=False, keep_init=False)
    if "discrete" in which:  # one decoder conditioned on an embedded generator id (standard_discrete.py)
        main("discrete_g2", ["--experiment", "discrete"], nan=False, keep_init=False)
    if "masked_sgan" in which:  # masked batch through the Social-GAN pooling (the K-times repeated scene list)
        main("masked_sgan_g2", ["--pool_type", "sgan"], nan=True, keep_init=False)
    if "sgan" in which:  # Social-GAN pooling in G and D (social_gan.py:157-229) instead of the attention
        main("pool_sgan_g2", ["--pool_type", "sgan"], nan=False, keep_init=False)
    if "narrow" in which:  # widths below the built ones (config.py:70-71): the HIP path runs them zero-padded (mggan/model/widths.py)
        main("narrow_h16_g2", ["--h_dim", "16", "--decoder_h_dim", "16"], nan=False, keep_init=True)
        main("narrow_h24_d8_g2", ["--h_dim", "24", "--decoder_h_dim", "8", "--noise_dim", "4"], nan=True, keep_init=True)
        main("narrow_sgan_g2", ["--pool_type", "sgan", "--h_dim", "16", "--decoder_h_dim", "24"], nan=False, keep_init=True)
        main("narrow_discrete_g2", ["--experiment", "discrete", "--h_dim", "24", "--decoder_h_dim", "16"], nan=False, keep_init=True)
