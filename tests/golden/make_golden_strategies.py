"""Golden vectors for the prediction strategies of the reference
(/root/reference/mggan/model/train.py:291-563: predict_expected, predict_uniform = 'uniform_expected' /
'smart_expected', predict_smart_sampling = 'smart_sampling' / 'uniform_sampling', predict_rejection),
produced by running the REAL reference on CPU from the initial state stored in golden_g{1,4}.npz.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_strategies.py
Writes tests/golden/golden_strategies_g{1,4}.npz (data only: recorded draws and expected outputs)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

ref_train, ref_config = _refload.load_reference()
import test_tube  # noqa: E402  (stub)
import mggan.model.modules.standard as ref_standard  # noqa: E402

np.int = int  # the reference still uses the alias numpy removed (train.py:310)


def t2n(t):
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.asarray(t)


def run(tag):
    g = dict(np.load(os.path.join(HERE, "golden_{}.npz".format(tag))))
    num_gens = int(g["meta/num_gens"])
    args = ref_config.get_parser().parse_args(["--gpus", "", "--num_gens", str(num_gens)])
    G, D = ref_train.construct_model(args)
    G.load_state_dict({k[3:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("G0/")})
    model = ref_train.PiNetMultiGeneratorGAN(G, D, args, test_tube.Experiment())
    # spread the PM-network output so that the strategies have something to rank (the initial logits are ~uniform)
    with torch.no_grad():
        G.net_chooser[4].bias.copy_(torch.linspace(-1.0, 1.0, num_gens))
    sub = [[int(s), int(e)] for s, e in g["meta/scenes"]]
    in_xy, in_dxdy, img = (torch.from_numpy(g["in/" + k].copy()) for k in ("in_xy", "in_dxdy", "features"))
    out = {"pm_bias": t2n(G.net_chooser[4].bias)}
    K = 20

    def noise(n, seed):
        torch.manual_seed(seed)
        return torch.stack([ref_standard.get_global_noise(8, sub, "gaussian") for _ in range(n)])

    n1 = noise(K, 101)
    pa, pr, probs, idx = model.predict_expected(in_dxdy, in_xy, sub, img=img, num=K, noise=n1)
    out.update({"expected/noise": t2n(n1), "expected/abs": t2n(pa), "expected/rel": t2n(pr), "expected/probs": t2n(probs),
                "expected/idx": t2n(idx)})
    n2 = noise(K * num_gens, 102)
    for name, eps in (("uniform_expected", 0.0), ("smart_expected", 1.0 / num_gens)):
        pa, pr, probs, idx = model.predict_uniform(in_dxdy, in_xy, sub, img=img, num=K, noise=n2, eps=eps)
        out.update({name + "/abs": t2n(pa), name + "/rel": t2n(pr), name + "/idx": t2n(idx), name + "/eps": np.float64(eps)})
    out["uniform/noise"] = t2n(n2)
    for name, eps in (("smart_sampling", 1.0 / num_gens ** 2), ("uniform_sampling", 0.0)):
        torch.manual_seed(103)  # Categorical(...).sample draws from torch's global CPU generator
        pa, pr, probs, idx = model.predict_smart_sampling(in_dxdy, in_xy, sub, img=img, num=K, noise=n2, eps=eps)
        out.update({name + "/abs": t2n(pa), name + "/idx": t2n(idx), name + "/eps": np.float64(eps)})
    if num_gens == 1:
        total = K + int(np.ceil((1 - 0.7) * K))
        n3 = noise(total, 104)
        torch.manual_seed(105)  # the N perturbations eps_i ~ randn(total, b, 8) * sigma^2 come from the global generator
        pa, pr, probs, idx = model.predict_rejection(in_dxdy, in_xy, sub, img=img, num=K, noise=n3, sigma=1e-3, N=4)
        out.update({"rejection/noise": t2n(n3), "rejection/abs": t2n(pa), "rejection/idx": t2n(idx)})
    path = os.path.join(HERE, "golden_strategies_{}.npz".format(tag))
    np.savez_compressed(path, **out)
    print(tag, sorted(out), os.path.getsize(path))


if __name__ == "__main__":
    torch.set_num_threads(1)
    run("g4")
    run("g1")
