"""Shared helpers for parity tests (tolerances of SURVEY A.12)."""
import numpy as np
import torch


def sd_from(golden, prefix):
    p = prefix + "/"
    return {k[len(p):]: torch.from_numpy(v.copy()) for k, v in golden.items() if k.startswith(p)}


def batch_from(golden, device="cpu"):
    b = {k: torch.from_numpy(golden["in/" + k].copy()).to(device)
         for k in ("in_xy", "in_dxdy", "gt_xy", "gt_dxdy", "features")}
    b["seq_start_end"] = [[int(s), int(e)] for s, e in golden["meta/scenes"]]
    return b


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_close(a, b, rtol=1e-3, atol=1e-5, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def assert_grad_close(a, b, what="", tol=1e-3):
    """per-tensor relL2 <= tol plus element-wise atol = tol*max|g| (A.12 iii)."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.abs(b).max())
    if scale == 0.0:
        assert float(np.abs(a).max()) <= 1e-12, what
        return
    assert rel_l2(a, b) <= tol, (what, rel_l2(a, b))
    assert float(np.abs(a - b).max()) <= tol * scale + 1e-12, (what, float(np.abs(a - b).max()), scale)


def check_param_grads(golden, prefix, module, tol=1e-3, grads=None):
    """Every parameter gradient stored under `prefix/` must match (A.12 iii).
    Tensors whose reference gradient is structurally zero up to round-off (e.g. a
    conv bias in front of a train-mode BatchNorm: < 1e-4 of the group's largest
    gradient) are only required to be equally negligible."""
    named = dict(module.named_parameters())
    get = (lambda k: named[k].grad) if grads is None else (lambda k: grads.get(k))
    keys = [k for k in named if "{}/{}".format(prefix, k) in golden]
    assert keys, prefix
    group = max(float(np.abs(golden["{}/{}".format(prefix, k)]).max()) for k in keys)
    for k in named:
        key = "{}/{}".format(prefix, k)
        g = get(k)
        if key not in golden:
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        assert g is not None, key
        ref = golden[key]
        if float(np.abs(ref).max()) < 1e-4 * group:
            assert float(g.abs().max()) <= 1e-3 * group, key
        else:
            assert_grad_close(g, ref, key, tol)
