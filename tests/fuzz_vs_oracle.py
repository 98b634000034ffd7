"""Fuzz of the HIP path against the CPU oracle: random ragged batches, generator / sample counts, loss masks and
variants (--gan_obj, --weighting_target, --pool_type sgan, --experiment discrete), one D+G+PM iteration each with
injected draws, every scratch buffer NaN-poisoned.  Test infrastructure (imports the oracle).
    python tests/fuzz_vs_oracle.py [seed] [cases]      (tests/test_fuzz.py runs a few cases)"""
import contextlib
import io
import os
import random
import sys
from collections import defaultdict

import numpy as np  # noqa: F401
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "mg-gan_amd"), os.path.join(R, "oracle")]


TOL = 1e-3  # losses rtol 1e-3, post-step parameters relL2 1e-3


def run_cases(seed=0, cases=12, verbose=True):
    import mggan_oracle as O
    from mggan.data_utils import synthetic
    from mggan.hip import functions as HF
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN
    from mggan.rng import ReplayRNG

    HF.poison_scratch(True)
    try:
        return _run(seed, cases, verbose, O, synthetic, Experiment, get_parser, construct_model, PiNetMultiGeneratorGAN,
                    ReplayRNG)
    finally:
        HF.poison_scratch(False)


def _run(seed, cases, verbose, O, synthetic, Experiment, get_parser, construct_model, PiNetMultiGeneratorGAN, ReplayRNG):
    torch.cuda.set_device(0)
    rnd = random.Random(seed)
    failed = []
    for case in range(cases):
        g = rnd.choice([1, 2, 3, 5, 8]); K = rnd.choice([1, 4, 20])
        sizes = [rnd.randint(1, 9) for _ in range(rnd.randint(1, 9))]
        if rnd.random() < 0.25:  # a wide batch: 32 or 64 scenes, up to 32 pedestrians each
            sizes = [rnd.randint(1, 32) for _ in range(rnd.choice([32, 64]))]
        extra = rnd.choice([[], [], ["--pool_type", "sgan"], ["--gan_obj", "LS"], ["--gan_obj", "MM"], ["--weighting_target", "l2"], ["--experiment", "discrete"]])
        nanmask = rnd.random() < 0.4
        cfg = get_parser().parse_args(["--num_gens", str(g), "--num_samples", str(K)] + extra)
        o = {extra[i].lstrip("-"): extra[i+1] for i in range(0, len(extra), 2)}
        torch.manual_seed(case + 100)
        with contextlib.redirect_stdout(io.StringIO()):
            G, D = construct_model(cfg)
        Go, Do = O.construct_oracle(g, gan_obj=o.get("gan_obj","NS"), pool_type=o.get("pool_type","sways"), experiment=o.get("experiment","multi_generator"))
        Go.load_state_dict(G.state_dict()); Do.load_state_dict(D.state_dict())
        tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
        tro = O.OracleTrainer(Go, Do, num_samples=K, gan_obj=o.get("gan_obj","NS"), weighting_target=o.get("weighting_target","ml"))
        batch = synthetic.make_batch(sizes, seed=case)
        b = sum(sizes); sc = batch["seq_start_end"]
        mask = torch.ones(b, dtype=torch.bool)
        if nanmask and b > 2:
            for p in rnd.sample(range(b), max(1, b // 4)):
                mask[p] = False
        dbatch = tr.to_device(batch)
        for M in (tr.G, tr.D, Go, Do): M.train()
        gen = torch.Generator().manual_seed(case)
        bm = int(mask.sum())
        cpu_args = (batch["in_xy"], batch["in_dxdy"], batch["gt_xy"][:, mask], batch["gt_dxdy"][:, mask], sc)
        gpu_args = (dbatch["in_xy"], dbatch["in_dxdy"], dbatch["gt_xy"][:, mask.cuda()], dbatch["gt_dxdy"][:, mask.cuda()], sc)
        ok = True; msg = ""
        try:
            for step, k in (("discriminator_step", 1), ("generator_step", K), ("net_chooser_step", 1)):
                noise = torch.randn(k, len(sc), 8, generator=gen).repeat_interleave(torch.tensor(sizes), dim=1)
                idx = torch.randint(0, g, (bm, k), generator=gen)
                labels = [(0.95, 0.05), (0.93, 0.07)]
                draws = {"noise": noise, "gen_idxs": idx, "labels": labels[0], "labels1": labels[0], "labels2": labels[1]}
                tr.rng = tr.G.rng = ReplayRNG(labels=list(labels), noise=[noise], gen_idxs=[idx])
                m_gpu, m_cpu = defaultdict(list), defaultdict(list)
                mk = None if bool(mask.all()) else mask
                getattr(tr, step)(*gpu_args, m_gpu, None if mk is None else mk.cuda(), dbatch["features"])
                getattr(tro, step)(*cpu_args, m_cpu, mask, batch["features"], draws=draws)
                for key, v in m_cpu.items():
                    if "probs" in key: continue
                    if not abs(m_gpu[key][0] - v[0]) <= TOL * abs(v[0]) + 1e-5:
                        ok = False; msg += " %s:%s gpu=%.6g cpu=%.6g" % (step[:3], key.split("/")[-1], m_gpu[key][0], v[0])
            for name, mod, ref in (("G", tr.G, Go), ("D", tr.D, Do)):
                a = torch.cat([p.detach().cpu().flatten() for p in mod.parameters()]).double()
                r = torch.cat([p.detach().flatten() for p in ref.parameters()]).double()
                rel = float((a - r).norm() / r.norm())
                if not rel <= TOL: ok = False; msg += " %s rel=%.2e" % (name, rel)
        except Exception as e:
            ok = False; msg = " EXC %s: %s" % (type(e).__name__, str(e)[:150])
        if not ok:
            failed.append((case, msg))
        if verbose:
            print("case %2d g=%d K=%2d sizes=%s mask=%s %s -> %s%s" % (case, g, K, sizes, "part" if not bool(mask.all()) else "all", extra, "ok" if ok else "FAIL", msg), flush=True)
    return failed


if __name__ == "__main__":
    bad = run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    print("failures:", len(bad))
    sys.exit(1 if bad else 0)
