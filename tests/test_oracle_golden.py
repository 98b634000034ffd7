"""CPU: the oracle restatement vs golden vectors produced by the real reference."""
import copy
from collections import defaultdict

import numpy as np
import pytest
import torch

import mggan_oracle as O
from helpers import assert_close, assert_grad_close, batch_from, rel_l2, sd_from

torch.set_num_threads(4)


def build(golden, which="0"):
    g = int(golden["meta/num_gens"])
    G, D = O.construct_oracle(g)
    G.load_state_dict(sd_from(golden, "G" + which), strict=True)
    D.load_state_dict(sd_from(golden, "D" + which), strict=True)
    G.train()
    D.train()
    return G, D


def T(golden, k):
    return torch.from_numpy(golden[k].copy())


def check_grads(golden, prefix, module, tol=1e-3):
    from helpers import check_param_grads
    check_param_grads(golden, prefix, module, tol)


def test_state_dict_surface_and_seeded_init(golden):
    """Same keys/shapes as the reference, and the same construction order: one
    seed reproduces the reference's initial weights bit for bit (App. B)."""
    g = int(golden["meta/num_gens"])
    torch.manual_seed(int(golden["meta/seed"]))
    G, D = O.construct_oracle(g)
    for mod, pre in ((G, "G0"), (D, "D0")):
        ref = sd_from(golden, pre)
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ref.keys())
        for k in ref:
            assert sd[k].shape == ref[k].shape, k
            assert torch.equal(sd[k], ref[k]), k


@pytest.mark.parametrize("mode", ["block", "faithful"])
def test_unit_modules(golden, mode):
    G, D = build(golden)
    bt = batch_from(golden)
    sc = bt["seq_start_end"]
    for name, enc in (("u_genc", G.encoder), ("u_denc", D.in_encoder)):
        enc.zero_grad()
        y = enc.run(bt["in_dxdy"])
        assert_close(y, golden[name + "/out"], what=name)
        (y * T(golden, name + "/cot")).sum().backward()
        check_grads(golden, name + "/grad", enc)
    for name, mod in (("u_gscene", copy.deepcopy(G.scene_encoder)), ("u_dscene", copy.deepcopy(D.scene_encoder))):
        mod.train()
        y = mod.run(bt["features"], True)
        assert_close(y, golden[name + "/out"], what=name)
        (y * T(golden, name + "/cot")).sum().backward()
        check_grads(golden, name + "/grad", mod)
        for k, v in mod.state_dict().items():
            assert_close(v, golden["{}/after/{}".format(name, k)], what=name + k)
        assert_close(mod.run(bt["features"], False), golden[name + "/out_eval"], what=name + " eval")
    for name, mod in (("u_gsoc", G.social), ("u_dsoc", D.social)):
        mod.zero_grad()
        h = T(golden, name + "/h").requires_grad_()
        y = mod.run(bt["in_xy"][-1], bt["in_dxdy"][-1], h, sc, mode)
        assert_close(y, golden[name + "/out"], what=name)
        (y * T(golden, name + "/cot")).sum().backward()
        assert_grad_close(h.grad, golden[name + "/grad_h"], name + " dh")
        check_grads(golden, name + "/grad", mod)


def test_decoder_rollout(golden):
    G, _ = build(golden)
    bt = batch_from(golden)
    dec = G.gs[G.n_gs - 1]
    h0 = T(golden, "u_dec/h0").requires_grad_()
    soc = T(golden, "u_dec/soc").requires_grad_()
    pa, pr = dec.run(bt["in_xy"][-1].repeat(2, 1), bt["in_dxdy"][-1].repeat(2, 1), soc, h0, torch.zeros_like(h0))
    assert_close(pa, golden["u_dec/abs"], what="abs")
    assert_close(pr, golden["u_dec/rel"], what="rel")
    ((pa * T(golden, "u_dec/cot_abs")).sum() + (pr * T(golden, "u_dec/cot_rel")).sum()).backward()
    assert_grad_close(h0.grad, golden["u_dec/grad_h0"], "dh0")
    assert_grad_close(soc.grad, golden["u_dec/grad_soc"], "dsoc")
    check_grads(golden, "u_dec/grad", dec)


@pytest.mark.parametrize("mode", ["block", "faithful"])
def test_discriminator_forward_backward(golden, mode):
    _, D = build(golden)
    bt = batch_from(golden)
    pdx = T(golden, "u_D/pred_dxdy").requires_grad_()
    o, br = D(bt["in_xy"], bt["in_dxdy"], T(golden, "u_D/pred_xy"), pdx, bt["seq_start_end"], img=bt["features"],
              mask=torch.ones(bt["in_xy"].shape[1], dtype=torch.bool), mode=mode)
    assert_close(o, golden["u_D/out"], what="out")
    assert_close(br, golden["u_D/branch"], what="branch")
    ((o * T(golden, "u_D/cot_out")).sum() + (br * T(golden, "u_D/cot_branch")).sum()).backward()
    assert_grad_close(pdx.grad, golden["u_D/grad_pred_dxdy"], "dpred")
    check_grads(golden, "u_D/grad", D)


@pytest.mark.parametrize("mode", ["block", "faithful"])
def test_generator_forward_backward(golden, mode):
    G, _ = build(golden)
    bt = batch_from(golden)
    idx = T(golden, "u_G/gen_idxs")
    K = idx.shape[1]
    go, logits, gi = G(bt["in_xy"], bt["in_dxdy"], bt["seq_start_end"], noise=T(golden, "u_G/noise"),
                       all_gen_out=False, img=bt["features"], num_samples=K, mode=mode, gen_idxs=idx)
    assert_close(logits, golden["u_G/logits"], what="logits")
    assert_close(go.abs, golden["u_G/abs"], what="abs")
    assert_close(go.rel, golden["u_G/rel"], what="rel")
    ((go.abs * T(golden, "u_G/cot_abs")).sum() + (go.rel * T(golden, "u_G/cot_rel")).sum()).backward()
    check_grads(golden, "u_G/grad", G)

    G, _ = build(golden)
    E = golden["u_Gall/noise"].shape[0]
    go, logits, _ = G(bt["in_xy"], bt["in_dxdy"], bt["seq_start_end"], noise=T(golden, "u_Gall/noise"),
                      all_gen_out=True, img=bt["features"], num_samples=E, mode=mode)
    assert_close(go.abs, golden["u_Gall/abs"], what="abs all")
    assert_close(go.rel, golden["u_Gall/rel"], what="rel all")
    assert_close(logits, golden["u_Gall/logits"], what="logits all")
    (logits * T(golden, "u_Gall/cot_logits")).sum().backward()
    check_grads(golden, "u_Gall/grad", G)


def _draws(golden, p, step):
    lab = golden.get(p + "/labels")
    d = {"noise": T(golden, p + "/noise"), "gen_idxs": T(golden, p + "/gen_idxs")}
    if step == "d":
        d["labels1"], d["labels2"] = tuple(lab[0]), tuple(lab[1])
    elif step == "g":
        d["labels"] = tuple(lab[0])
    return d


def _run_iterations(golden, mode, teacher_forced):
    G, D = build(golden)
    tr = O.OracleTrainer(G, D, mode=mode)
    bt = batch_from(golden)
    args = (bt["in_xy"], bt["in_dxdy"], bt["gt_xy"], bt["gt_dxdy"], bt["seq_start_end"])
    mask = torch.ones(bt["in_xy"].shape[1], dtype=torch.bool)
    for it in range(1, 4):
        for step, fn, mod in (("d", tr.discriminator_step, D), ("g", tr.generator_step, G),
                              ("pm", tr.net_chooser_step, G)):
            p = "s{}_{}".format(it, step)
            m = defaultdict(list)
            fn(*args, m, mask, bt["features"], draws=_draws(golden, p, step))
            for k, v in m.items():
                ref = float(golden[p + "/metric/" + k])
                assert abs(v[0] - ref) <= 1e-3 * abs(ref) + 1e-6, (p, k, v[0], ref)
            if it == 1:
                check_grads(golden, p + "/grad", mod)
        if it in (1, 3):
            for mod, pre in ((G, "G"), (D, "D")):
                ref = sd_from(golden, pre + str(it))
                sd = mod.state_dict()
                a = torch.cat([sd[k].flatten().double() for k in ref if ref[k].is_floating_point()])
                r = torch.cat([ref[k].flatten().double() for k in ref if ref[k].is_floating_point()])
                assert rel_l2(a.numpy(), r.numpy()) <= 1e-3, (pre, it, rel_l2(a.numpy(), r.numpy()))
                for k in ref:
                    if not ref[k].is_floating_point():
                        assert int(sd[k]) == int(ref[k]), k
    return G, D


@pytest.mark.parametrize("mode", ["block", "faithful"])
def test_three_training_iterations(golden, mode):
    """D/G/PM steps with the recorded draws: every logged loss (rtol 1e-3), every
    parameter gradient of iteration 1 (relL2 1e-3), parameters after iterations 1
    and 3 (relL2 1e-3), BN num_batches_tracked exactly."""
    _run_iterations(golden, mode, False)


def test_rng_draw_order_matches_reference(golden):
    """Without injection the oracle draws noise/labels itself in the reference's
    order (App. B): seeding like the fixture must reproduce the recorded draws."""
    G, D = build(golden)
    tr = O.OracleTrainer(G, D)
    bt = batch_from(golden)
    args = (bt["in_xy"], bt["in_dxdy"], bt["gt_xy"], bt["gt_dxdy"], bt["seq_start_end"])
    mask = torch.ones(bt["in_xy"].shape[1], dtype=torch.bool)
    for step, fn in (("d", tr.discriminator_step), ("g", tr.generator_step), ("pm", tr.net_chooser_step)):
        p = "s1_" + step
        torch.manual_seed(int(golden[p + "/seed_torch"]))
        np.random.seed(int(golden[p + "/seed_numpy"]))
        m = defaultdict(list)
        r = fn(*args, m, mask, bt["features"])
        if step != "pm":
            assert torch.equal(r["gen_idxs"], T(golden, p + "/gen_idxs")), step
        for k, v in m.items():
            ref = float(golden[p + "/metric/" + k])
            assert abs(v[0] - ref) <= 1e-3 * abs(ref) + 1e-6, (p, k, v[0], ref)


def test_predict_and_ade_fde(golden):
    G, _ = build(golden, "3")
    G.eval()
    bt = batch_from(golden)
    with torch.no_grad():
        go, logits, _ = G(bt["in_xy"], bt["in_dxdy"], bt["seq_start_end"], noise=T(golden, "e/noise"),
                          all_gen_out=False, img=bt["features"], num_samples=20,
                          gen_idxs=torch.from_numpy(golden["e/gen_idxs"].copy()))
    assert_close(go.abs, golden["e/abs"], what="predict")
    assert_close(torch.softmax(logits, 1), golden["e/probs"], what="probs")
    m = O.compute_metrics(go.abs, bt["gt_xy"], bt["seq_start_end"])
    for k in ("ADE", "FDE", "Mode"):
        np.testing.assert_allclose(m[k], golden["e/" + k], rtol=1e-3)
    # the reference's own numbers on its own predictions (pure metric check)
    m = O.compute_metrics(T(golden, "e/abs"), bt["gt_xy"], bt["seq_start_end"])
    for k in ("ADE", "FDE", "Mode"):
        np.testing.assert_allclose(m[k], golden["e/" + k], rtol=1e-6)
