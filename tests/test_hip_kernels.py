"""GPU: C-ABI primitives (generic GEMM family, reductions, optimizer) against plain torch math."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda")


def _lib():
    from mggan.hip import lib
    return lib


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rows,K,N,act", [(7, 3, 32, 1), (300, 136, 32, 0), (1000, 192, 96, 1), (65, 96, 1, 3),
                                          (129, 16, 4, 0), (64, 24, 64, 2)])
def test_linear_fwd_bwd_wgrad(rows, K, N, act):
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(rows + K)
    ldx = K + 5
    Xf = torch.randn(rows, ldx, generator=g)
    W, bias = torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
    X = Xf[:, :K]
    slope = 0.2
    pre = X.double() @ W.double().t() + bias.double()
    ref = {0: pre, 1: torch.where(pre > 0, pre, pre * slope), 2: torch.sigmoid(pre),
           3: torch.sigmoid(pre) * (1 - 2e-7) + 1e-7}[act]
    Xd, Wd, bd = Xf.to(dev), W.to(dev), bias.to(dev)
    Y = torch.empty(rows, N, device=dev)
    lib.mggan_linear_fwd(Xd.data_ptr(), ldx, Wd.data_ptr(), bd.data_ptr(), Y.data_ptr(), N, rows, K, N, act, slope, st())
    np.testing.assert_allclose(Y.cpu().numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)
    dY = torch.randn(rows, N, generator=g)
    yr = ref.float()
    dact = {0: torch.ones_like(yr), 1: torch.where(yr > 0, 1.0, slope), 2: yr * (1 - yr), 3: yr * (1 - yr)}[act]
    dZ_ref = dY * dact
    dZ = torch.empty(rows, N, device=dev)
    dYd = dY.to(dev)
    lib.mggan_act_bwd(dYd.data_ptr(), N, Y.data_ptr(), N, dZ.data_ptr(), N, rows, N, act, slope, st())
    np.testing.assert_allclose(dZ.cpu().numpy(), dZ_ref.numpy(), rtol=1e-4, atol=1e-6)
    dX = torch.full((rows, K), 1.0, device=dev)
    lib.mggan_linear_bwd_data(dZ.data_ptr(), N, Wd.data_ptr(), K, dX.data_ptr(), K, rows, K, N, 1, 0, 0, 0, 0.0, st())
    np.testing.assert_allclose(dX.cpu().numpy(), (dZ_ref.double() @ W.double()).float().numpy() + 1.0, rtol=1e-4,
                               atol=1e-4)
    dW = torch.full((N, K), 0.5, device=dev)
    db = torch.zeros(N, device=dev)
    nb = lib.mggan_wgrad_workspace_bytes(rows, K, N, 0)
    ws = torch.empty(nb // 4, device=dev)
    lib.mggan_wgrad(dZ.data_ptr(), N, Xd.data_ptr(), ldx, dW.data_ptr(), K, db.data_ptr(), rows, K, N, 0, 1, 0, 0, 0,
                    0, 0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    dW_ref = dZ_ref.double().t() @ X.double()
    np.testing.assert_allclose(dW.cpu().numpy(), dW_ref.float().numpy() + 0.5, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), dZ_ref.double().sum(0).float().numpy(), rtol=1e-4, atol=1e-4)
    # fused activation derivative (dY and Y in, no act_bwd launch) gives the same input / weight gradients
    if act != 0:
        dX3 = torch.zeros(rows, K, device=dev)
        lib.mggan_linear_bwd_data(dYd.data_ptr(), N, Wd.data_ptr(), K, dX3.data_ptr(), K, rows, K, N, 0, Y.data_ptr(), N,
                                  act, slope, st())
        np.testing.assert_allclose(dX3.cpu().numpy(), (dZ_ref.double() @ W.double()).float().numpy(), rtol=1e-4, atol=1e-4)
        dW3, db3 = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        lib.mggan_wgrad(dYd.data_ptr(), N, Xd.data_ptr(), ldx, dW3.data_ptr(), K, db3.data_ptr(), rows, K, N, 0, 1, 0, 0,
                        0, 0, Y.data_ptr(), N, act, slope, ws.data_ptr(), nb, st())
        np.testing.assert_allclose(dW3.cpu().numpy(), dW_ref.float().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(db3.cpu().numpy(), dZ_ref.double().sum(0).float().numpy(), rtol=1e-4, atol=1e-4)
    # feature-major operands ([feature][row]) give the same result
    dZt, Xt = dZ.t().contiguous(), Xd[:, :K].t().contiguous()
    dW2, db2 = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    lib.mggan_wgrad(dZt.data_ptr(), rows, Xt.data_ptr(), rows, dW2.data_ptr(), K, db2.data_ptr(), rows, K, N, 0, 1, 0, 0,
                    0, 1, 0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    np.testing.assert_allclose(dW2.cpu().numpy(), dW_ref.float().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db2.cpu().numpy(), dZ_ref.double().sum(0).float().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,ld,K,N", [(25600, 25600, 32, 64), (1003, 1003, 3, 32), (4099, 4100, 8, 32),
                                         (100003, 100004, 32, 16), (70000, 70000, 64, 64), (15, 16, 16, 32),
                                         (5000, 5000, 33, 17)])
def test_wgrad_streaming_feature_major(rows, ld, K, N):
    """The streaming kernel behind feature-major weight gradients (<= 64 x 64 outputs): aligned and unaligned row
    strides, slabs with partial super-steps, every tile-count variant; against f64."""
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(rows + K + N)
    dZ = torch.randn(N, ld, generator=g)
    X = torch.randn(K, ld, generator=g)
    dZd, Xd = dZ.to(dev), X.to(dev)
    dW, db = torch.full((N, K), 0.25, device=dev), torch.full((N,), -1.0, device=dev)
    nb = lib.mggan_wgrad_workspace_bytes(rows, K, N, 0)
    assert nb == lib.mggan_wgrad_splits(rows, K, N, 0) * N * (K + 1) * 4
    ws = torch.full((nb // 4,), float("nan"), device=dev)
    lib.mggan_wgrad(dZd.data_ptr(), ld, Xd.data_ptr(), ld, dW.data_ptr(), K, db.data_ptr(), rows, K, N, 0, 1, 0, 0, 0, 1,
                    0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    ref = dZ[:, :rows].double() @ X[:, :rows].double().t()
    scale = float(rows) ** 0.5
    np.testing.assert_allclose(dW.cpu().numpy(), ref.float().numpy() + 0.25, rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dZ[:, :rows].double().sum(1).float().numpy() - 1.0, rtol=1e-4,
                               atol=2e-5 * scale)
    # bit-reproducible (fixed summation order)
    dW2, db2 = torch.full((N, K), 0.25, device=dev), torch.full((N,), -1.0, device=dev)
    lib.mggan_wgrad(dZd.data_ptr(), ld, Xd.data_ptr(), ld, dW2.data_ptr(), K, db2.data_ptr(), rows, K, N, 0, 1, 0, 0, 0, 1,
                    0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("rows,K,N", [(5003, 64, 256), (20000, 136, 32), (16384, 192, 96), (333, 1, 64), (4096, 65, 65),
                                      (131, 24, 64)])
def test_wgrad_streaming_row_major(rows, K, N):
    """Row-major operands through the streaming kernel: outputs wider than one 64 x 64 panel, padded row strides."""
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(rows + K + N)
    ldz, ldx = N + 3, K + 1
    dZ, X = torch.randn(rows, ldz, generator=g), torch.randn(rows, ldx, generator=g)
    dZd, Xd = dZ.to(dev), X.to(dev)
    dW, db = torch.full((N, K), 0.25, device=dev), torch.full((N,), -1.0, device=dev)
    nb = lib.mggan_wgrad_workspace_bytes(rows, K, N, 0)
    ws = torch.full((nb // 4,), float("nan"), device=dev)
    lib.mggan_wgrad(dZd.data_ptr(), ldz, Xd.data_ptr(), ldx, dW.data_ptr(), K, db.data_ptr(), rows, K, N, 0, 1, 0, 0, 0, 0,
                    0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    ref = dZ[:, :N].double().t() @ X[:, :K].double()
    scale = float(rows) ** 0.5
    np.testing.assert_allclose(dW.cpu().numpy(), ref.float().numpy() + 0.25, rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dZ[:, :N].double().sum(0).float().numpy() - 1.0, rtol=1e-4,
                               atol=2e-5 * scale)


def test_wgrad_grouped_segments():
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(3)
    T, K, N, ng = 12, 32, 128, 4
    seg = torch.tensor([0, 5, 5, 40, 97], dtype=torch.int32)  # one empty group
    R = int(seg[-1])
    dZ, X = torch.randn(R * T, N, generator=g), torch.randn(R * T, K, generator=g)
    stride = N * K + 64
    dW = torch.zeros(ng * stride, device=dev)
    db = torch.zeros(ng * stride, device=dev)
    nb = lib.mggan_wgrad_workspace_bytes(R * T, K, N, ng)
    ws = torch.empty(nb // 4, device=dev)
    dZd, Xd, segd = dZ.to(dev), X.to(dev), seg.to(dev)
    lib.mggan_wgrad(dZd.data_ptr(), N, Xd.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), R * T, K, N,
                    segd.data_ptr(), T, ng, stride, stride, 0, 0, 0, 0, 0.0, ws.data_ptr(), nb, st())
    for gi in range(ng):
        a, b = int(seg[gi]) * T, int(seg[gi + 1]) * T
        ref = dZ[a:b].double().t() @ X[a:b].double()
        np.testing.assert_allclose(dW[gi * stride:gi * stride + N * K].view(N, K).cpu().numpy(), ref.float().numpy(),
                                   rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(db[gi * stride:gi * stride + N].cpu().numpy(), dZ[a:b].double().sum(0).float().numpy(),
                                   rtol=1e-4, atol=2e-4)


def test_grad_reduce_multi_layouts():
    """The batched fixed-order reduction on every layout its producers leave behind: 16-byte aligned partial blocks (one
    vector load per lane), unaligned bases / odd pitches / odd totals (scalar loads), groups, the bias column, storing
    and accumulating destinations, 1 ... 1,030 partial blocks, outputs narrower and wider than a 256-float chunk; a second
    launch on the same inputs gives the same bits."""
    import ctypes
    from mggan.hip import functions as HF

    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(11)
    cases = [  # (M, Naug, has_bias, splits, groups, pitch slack, base offset in floats, overwrite)
        (128, 32, 0, 64, 8, 948, 0, False), (128, 2, 0, 64, 8, 4788, 1, True), (16, 33, 1, 81, 8, 0, 0, False),
        (96, 193, 1, 29, 1, 0, 0, False), (1, 97, 1, 128, 1, 0, 0, False), (256, 65, 1, 128, 1, 0, 0, True),
        (1, 2304, 0, 512, 1, 16, 0, False), (32, 17, 1, 1030, 1, 528, 0, False), (3, 5, 1, 1, 2, 1, 3, False),
        (64, 4, 0, 7, 3, 0, 2, False)]
    descs, checks, keep = [], [], []
    for M, Naug, hb, splits, groups, slack, off, ow in cases:
        total = M * Naug
        pitch = total + slack
        P = torch.randn(off + groups * splits * pitch, generator=g).to(dev)
        N = Naug - 1 if hb else Naug
        lddw = N + 2
        w_stride, b_stride = M * lddw + 5, M + 3
        dW = torch.full((groups * w_stride,), 0.5, device=dev)
        db = torch.full((groups * b_stride,), -0.25, device=dev)
        descs.append(HF._ReduceDesc(P.data_ptr() + 4 * off, dW.data_ptr(), db.data_ptr() if hb else None, w_stride, b_stride, M,
                                    Naug, hb | (2 if ow else 0), lddw, splits, groups, pitch, 0))
        ref = P[off:].view(groups, splits, pitch)[:, :, :total].double().sum(1).view(groups, M, Naug).cpu()
        checks.append((dW, db, ref, M, N, hb, lddw, w_stride, b_stride, groups, ow, splits))
        keep.append(P)
    arr = (HF._ReduceDesc * len(descs))(*descs)

    def run():
        for dW, db, *_ in checks:
            dW.fill_(0.5)
            db.fill_(-0.25)
        lib.mggan_grad_reduce_multi(ctypes.addressof(arr), len(descs), st())
        torch.cuda.synchronize()
        return [(c[0].clone(), c[1].clone()) for c in checks]

    first = run()
    for (dW, db, ref, M, N, hb, lddw, w_stride, b_stride, groups, ow, splits), (w1, b1) in zip(checks, first):
        base = 0.0 if ow else 0.5
        tol = 3e-6 * splits ** 0.5 + 1e-6
        for gi in range(groups):
            got = dW[gi * w_stride:gi * w_stride + M * lddw].view(M, lddw).cpu().double()
            np.testing.assert_allclose(got[:, :N].numpy(), ref[gi, :, :N].numpy() + base, rtol=1e-5, atol=tol)
            assert torch.all(got[:, N:] == 0.5)  # the padding columns of the destination are not touched
            if hb:
                gb = db[gi * b_stride:gi * b_stride + M].cpu().double()
                np.testing.assert_allclose(gb.numpy(), ref[gi, :, N].numpy() + (0.0 if ow else -0.25), rtol=1e-5, atol=tol)
    for (w1, b1), (w2, b2) in zip(first, run()):
        assert torch.equal(w1, w2) and torch.equal(b1, b2)


def test_gather_sum_and_transpose():
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(5)
    b, K, C = 13, 6, 40
    src = torch.randn(K * b, C, generator=g)
    inv = torch.randperm(K * b, generator=g).to(torch.int32)
    dst = torch.zeros(b, C, device=dev)
    srcd, invd = src.to(dev), inv.to(dev)
    lib.mggan_gather_sum(srcd.data_ptr(), C, invd.data_ptr(), dst.data_ptr(), C, b, K, C, 0, st())
    ref = src[inv.long()].view(K, b, C).sum(0)
    np.testing.assert_allclose(dst.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    W = torch.randn(32, 136, generator=g)
    WT = torch.empty(136, 32, device=dev)
    Wd = W.to(dev)
    lib.mggan_transpose(Wd.data_ptr(), WT.data_ptr(), 32, 136, st())
    assert torch.equal(WT.cpu(), W.t().contiguous())


def test_clip_adamw_matches_torch_and_skips_untouched():
    """Fused clip+AdamW vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over 3 steps; a parameter
    without gradient is left bit-identical."""
    from mggan.hip.flat import FlatModule
    from mggan.optim import FlatAdamW

    dev = _dev()

    class M(FlatModule):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(7, 5)
            self.b = torch.nn.Linear(5, 3)
            self.c = torch.nn.Parameter(torch.randn(11))

    torch.manual_seed(0)
    m = M().to(dev).flatten_parameters_()
    ref = M()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    opt = FlatAdamW(m, lr=1e-2, betas=(0.5, 0.999))
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(1)
    for it in range(3):
        opt.zero_grad()
        ropt.zero_grad()
        for (n, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
            if n == "c" and it != 1:
                continue  # untouched in steps 0 and 2
            gr = torch.randn(p.shape, generator=g) * (50.0 if it == 0 else 0.1)
            ptr = m.grad_ptr(p)
            p.grad.copy_(gr.to(dev))
            rp.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        ropt.step()
        opt.step(10.0)
        for (n, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), rp.detach().numpy(), rtol=2e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("rows,dims,acts,train_w,need_dx", [
    (1000, (192, 96, 1), ((1, 0.2), (3, 0.0)), True, True),     # discriminator head (sigmoid-eps output)
    (77, (192, 96, 4), ((1, 0.2), (0, 0.0)), False, True),       # generator-id head, frozen weights (G step)
    (1280, (24, 64, 32), ((1, 0.2), (0, 0.0)), True, False),     # pred_encoder on leaf inputs
    (33, (128, 16, 16, 8), ((1, 0.0), (1, 0.0), (0, 0.0)), True, True),  # PM-network (3 layers, ReLU)
    (1, (65, 64, 33), ((0, 0.0), (2, 0.0)), True, True),         # odd widths, a single row
])
def test_mlp_chain_matches_torch(rows, dims, acts, train_w, need_dx):
    """mggan_mlp_chain (forward and backward chains) == the same nn.Sequential in float64 on the CPU and the
    one-layer GEMM launches it replaces."""
    import torch.nn as nn
    from mggan.hip import functions as HF
    from mggan.hip.flat import FlatModule

    dev = _dev()
    torch.manual_seed(rows + sum(dims))

    class Net(FlatModule):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

    net = Net()
    ref = [nn.Linear(a, b).double() for a, b in zip(dims[:-1], dims[1:])]
    for r, l in zip(ref, net.layers):
        r.load_state_dict({k: v.double() for k, v in l.state_dict().items()})
    net = net.to(dev).flatten_parameters_()
    for p in net.parameters():
        p.requires_grad_(train_w)
    x = torch.randn(rows, dims[0] + 3)[:, :dims[0]]  # strided rows
    xd = x.to(dev).requires_grad_(need_dx)
    xr = x.double().requires_grad_(True)

    def act_ref(v, a, s):
        return {0: v, 1: torch.where(v > 0, v, v * s), 2: torch.sigmoid(v), 3: torch.sigmoid(v) * (1 - 2e-7) + 1e-7}[a]

    h = xr
    for l, (a, s) in zip(ref, acts):
        h = act_ref(l(h), a, s)
    layers = [(l, a, s) for l, (a, s) in zip(net.layers, acts)]
    y = HF.mlp(xd, layers)
    np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().float().numpy(), rtol=2e-5, atol=2e-5)
    y1 = xd.detach()
    for l, a, s in layers:  # the unfused launches: same products, k summed in a different order inside 16-wide steps
        y1 = HF.linear(y1, l, a, s)
    np.testing.assert_allclose(y1.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    if not (train_w or need_dx):
        return
    dy = torch.randn(rows, dims[-1])
    h.backward(dy.double())
    net.zero_grad_flat()
    HF.defer_grad_reduce(False)
    y.backward(dy.to(dev))
    if need_dx:
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.float().numpy(), rtol=1e-4, atol=1e-5)
    for l, r in zip(net.layers, ref):
        if train_w:
            np.testing.assert_allclose(l.weight.grad.cpu().numpy(), r.weight.grad.float().numpy(), rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(l.bias.grad.cpu().numpy(), r.bias.grad.float().numpy(), rtol=1e-4, atol=2e-5)
        else:
            assert l.weight.grad is None or float(l.weight.grad.abs().max()) == 0.0


@pytest.mark.parametrize("H,sizes", [(32, [1, 2, 20, 1, 1, 7, 33, 64, 3, 5, 1]), (64, [20] * 7 + [1] * 70 + [13, 2]),
                                     (32, [1, 1, 1]), (64, [32] * 9 + [17, 16, 15]), (32, [16, 5, 4, 3, 2, 1] * 3),
                                     (64, [48, 64, 33, 1, 50]), (32, [20] * 300),
                                     (64, [24, 31, 17, 2]), (32, [32] * 4 + [19])])  # (few scenes of two blocks: row splits)
def test_social_attention_rows_match_the_unfused_launches(H, sizes):
    """The row-structured social attention (csrc/social_rows.hip: one launch per direction, MFMA pair MLP, nothing saved
    per pair, weight gradients inside the backward launch) against the per-stage entry points on ragged scenes: lone
    pedestrians, one to four neighbour blocks per row, scenes that fill 64, more scenes than workgroups."""
    from mggan.hip import functions as HF
    from mggan.model.modules.social import SocialAttention

    dev = _dev()
    torch.manual_seed(5)
    mod = SocialAttention(16, H).to(dev)
    mod.flatten_parameters_()
    b = sum(sizes)
    sse, s = [], 0
    for n in sizes:
        sse.append([s, s + n])
        s += n
    xy = torch.randn(8, b, 2, device=dev) * 3
    dxy = torch.randn(7, b, 2, device=dev)
    h0 = torch.randn(b, H, device=dev)
    cot = torch.randn(b, H, device=dev)
    tb = HF.scene_tables(sse, b, dev)
    assert tb.rows_ok and tb.max_n == max(sizes)
    res = []
    for fused in (True, False):
        tb.rows_ok = fused
        HF.poison_scratch(True)
        try:
            mod.zero_grad()
            h = h0.clone().requires_grad_()
            y = mod(xy, dxy, h, sse)
            (y * cot).sum().backward()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), h.grad.clone(), [p.grad.clone() for p in mod.parameters()]))
        finally:
            tb.rows_ok = True
            HF.poison_scratch(False)
    (y1, g1, p1), (y0, g0, p0) = res
    # same math, other summation orders (MFMA k order, shuffle trees): round-off apart
    torch.testing.assert_close(y1, y0, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-5)
    for (name, _), a, c in zip(mod.named_parameters(), p1, p0):
        scale = float(c.abs().max()) + 1e-12
        assert float((a - c).abs().max()) <= 2e-4 * scale + 1e-6, (name, float((a - c).abs().max()), scale)


def test_social_attention_rows_frozen_weights_and_pair_pass_positions():
    """The data-only backward (the generator step: discriminator frozen) and xy_mod (positions shared by the real and the
    fake half of a pair pass) of the row-structured kernels against the per-stage ones."""
    from mggan.hip import functions as HF
    from mggan.model.modules.social import SocialAttention

    dev = _dev()
    torch.manual_seed(11)
    H, sizes = 64, [20, 3, 1, 32, 8]
    mod = SocialAttention(16, H).to(dev)
    mod.flatten_parameters_()
    b = sum(sizes)
    sse, s = [], 0
    for n in sizes:
        sse.append([s, s + n])
        s += n
    sse2 = sse + [[s0 + b, e0 + b] for s0, e0 in sse]
    xy, dxy = torch.randn(8, b, 2, device=dev) * 3, torch.randn(7, b, 2, device=dev)
    h0, cot = torch.randn(2 * b, H, device=dev), torch.randn(2 * b, H, device=dev)
    tb = HF.scene_tables(sse2, 2 * b, dev)
    res = []
    for fused in (True, False):
        tb.rows_ok = fused
        try:
            for p in mod.parameters():
                p.requires_grad_(False)
            h = h0.clone().requires_grad_()
            y = mod(xy, dxy, h, sse2, xy_mod=b)
            (y * cot).sum().backward()
            res.append((y.detach().clone(), h.grad.clone()))
        finally:
            tb.rows_ok = True
            for p in mod.parameters():
                p.requires_grad_(True)
    torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("H,E,b", [(64, 64, 4100), (32, 16, 4133)])
def test_lstm_encoder_large_batch_matches_torch_f64(H, E, b):
    """The matrix-core encoder kernels (forward, and BPTT from 4,096 trajectories up) on a ragged row count: output and
    every parameter gradient against nn.Linear + nn.LSTM evaluated in float64 (common_modules.py:48-66)."""
    from mggan.model.modules.common_modules import TrajectoryEncoder

    torch.manual_seed(H + b)
    enc = TrajectoryEncoder(hidden_size=H, embedding_dim=E)
    ref_emb, ref_lstm = torch.nn.Linear(2, E).double(), torch.nn.LSTM(E, H).double()
    ref_emb.load_state_dict({k: v.double() for k, v in enc.embedding.state_dict().items()})
    ref_lstm.load_state_dict({k: v.double() for k, v in enc.encoder.state_dict().items()})
    enc = enc.to(_dev()).flatten_parameters_()
    x = torch.randn(7, b, 2) * 0.5
    cot = torch.randn(b, H)
    y = enc(x.to(_dev()))
    (y * cot.to(_dev())).sum().backward()
    _, (h, _) = ref_lstm(ref_emb(x.double()))
    (h[0] * cot.double()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), h[0].detach().float().numpy(), rtol=1e-4, atol=2e-6)
    got = dict(enc.named_parameters())
    for name, p in list(ref_emb.named_parameters(prefix="embedding")) + list(ref_lstm.named_parameters(prefix="encoder")):
        g, r = got[name].grad.detach().cpu().double(), p.grad
        assert float((g - r).norm() / r.norm()) < 1e-4, name


@pytest.mark.parametrize("g,sizes,K", [(4, [20] * 103 + [7, 1, 13], 4), (8, [32] * 40, 20)])
def test_discriminator_lean_heads_match_the_full_width_pass(monkeypatch, g, sizes, K):
    """K-sample pass of a frozen discriminator (the generator step): the lean heads (per-pedestrian part P + the pred_enc
    product per row, csrc/dheads.hip) against the full-width kernels over the assembled rows -- scores, generator-id
    logits and the gradient that flows back into the predictions.  b is not a multiple of 16 in the first case (the
    16-row tiles wrap around the pedestrians of a block)."""
    from mggan.data_utils.synthetic import make_batch
    from mggan.hip import functions as HF
    from mggan.hip.lib import load
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model

    dev = _dev()
    torch.manual_seed(21)
    _, D = construct_model(get_parser().parse_args(["--num_gens", str(g)]))
    D = D.to(dev).flatten_parameters_()
    D.train()
    for p in D.parameters():
        p.requires_grad_(False)
    bt = make_batch(sizes, seed=4, device=dev)
    b = bt["in_xy"].shape[1]
    monkeypatch.setattr(HF, "DHEADS_MIN_ROWS", 1024)
    pred_d0 = (bt["gt_dxdy"][:, None] + 0.3 * torch.randn(12, K, b, 2, device=dev)).contiguous()
    cot_o, cot_b = torch.randn(b, K, device=dev), torch.randn(b, K, g, device=dev)
    res = []
    # the one-launch row pass (steps -> heads), the lean heads over the assembled pred_enc block, the full-width pass
    for mode, (rows_lean, heads_lean) in enumerate((("1", "1"), ("0", "1"), ("0", "0"))):
        monkeypatch.setenv("MGGAN_DROWS_LEAN", rows_lean)
        monkeypatch.setenv("MGGAN_DHEADS_LEAN", heads_lean)
        pd = pred_d0.clone().requires_grad_()
        px = bt["in_xy"][-1][None, None] + torch.cumsum(pd, 0)
        calls = []
        L = load()
        L.trace = calls
        try:
            o, br = D(bt["in_xy"], bt["in_dxdy"], px, pd, bt["seq_start_end"], img=bt["features"])
            ((o * cot_o).sum() + (br * cot_b).sum()).backward()
        finally:
            L.trace = None
        names = {c[0] for c in calls}
        for entry, want in (("mggan_d_rows_lean_fwd", mode == 0), ("mggan_d_rows_lean_bwd", mode == 0),
                            ("mggan_dheads_lean_fwd", mode == 1), ("mggan_dheads_lean_bwd", mode == 1)):
            assert (entry in names) == want, (mode, entry)
        res.append((o.detach().clone(), br.detach().clone(), pd.grad.clone()))
    scale = float(res[2][2].abs().max())
    for k in (0, 1):
        torch.testing.assert_close(res[k][0], res[2][0], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(res[k][1], res[2][1], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(res[k][2], res[2][2], rtol=1e-4, atol=1e-5 * scale)


def test_decoder_h0_shared_part_matches_the_per_row_product(monkeypatch):
    """h0 = W_e2d [enc_h | noise] + b with the enc_h part computed once per pedestrian (mggan_decoder_e2d_shared + the
    noise columns per row; backward: dH0 folded over a pedestrian's rows, one product per pedestrian) against the
    per-row product: rollouts and every generator gradient (W_e2d / b_e2d included)."""
    from mggan.data_utils.synthetic import make_batch
    from mggan.hip import functions as HF
    from mggan.hip.lib import load
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.rng import ReplayRNG

    dev = _dev()
    g, K, sizes = 3, 6, [5, 1, 9, 17, 3, 32, 2]
    bt = make_batch(sizes, seed=9, device=dev)
    b = bt["in_xy"].shape[1]
    gen = torch.Generator().manual_seed(77)
    idx = torch.randint(0, g, (b, K), generator=gen)
    noise = torch.randn(K, len(sizes), 8, generator=gen).repeat_interleave(torch.tensor(sizes), dim=1).to(dev)
    cot = torch.randn(12, K, b, 2, generator=gen).to(dev)
    res = []
    for min_rows in (0, 1 << 30):
        monkeypatch.setattr(HF, "E2D_SHARED_MIN_ROWS", min_rows)
        torch.manual_seed(5)
        G, _ = construct_model(get_parser().parse_args(["--num_gens", str(g)]))
        G = G.to(dev).flatten_parameters_()
        G.train()
        G.rng = ReplayRNG(gen_idxs=[idx])
        calls = []
        L = load()
        L.trace = calls
        try:
            go, _, _ = G(bt["in_xy"], bt["in_dxdy"], bt["seq_start_end"], noise=noise, all_gen_out=False, img=bt["features"],
                         num_samples=K)
            ((go.abs * cot).sum() + (go.rel * cot.flip(0)).sum()).backward()
            HF.join_side_stream()
            torch.cuda.synchronize()
        finally:
            L.trace = None
        assert ("mggan_decoder_e2d_shared" in {c[0] for c in calls}) == (min_rows == 0)
        res.append((go.abs.detach().clone(), go.rel.detach().clone(),
                    {n: q.grad.detach().clone() for n, q in G.named_parameters() if q.grad is not None}))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-5, atol=2e-6)
    assert res[0][2].keys() == res[1][2].keys() and any("enc_h_to_dec_h" in n for n in res[0][2])
    gmax = max(float(v.abs().max()) for v in res[1][2].values())
    for n, ref in res[1][2].items():
        # (the bias of a convolution in front of a BatchNorm has a zero gradient: rounding noise of ~1e-8 on both sides)
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        torch.testing.assert_close(res[0][2][n], ref, rtol=2e-4, atol=2e-5 * scale, msg=lambda m, n=n: n + ": " + m)


@pytest.mark.parametrize("b,K,EIN,S,Z", [(1280, 20, 128, 32, 8), (37, 3, 128, 32, 8), (409, 5, 96, 0, 8), (64, 1, 72, 32, 4),
                                         (10, 20, 50, 18, 8)])
def test_rollout_ped_adjoint_matches_the_three_launches(b, K, EIN, S, Z):
    """mggan_rollout_ped_adjoint (the per-pedestrian tail of the rollout adjoint in one launch) against gather_sum ->
    linear_bwd_data -> gather_sum(accumulate) on the same operands: widths that are not multiples of 32, a social block that
    does not start on a 32-column boundary, no social block, pedestrian counts that do not fill a workgroup."""
    from mggan.hip import functions as HF
    from mggan.hip import lib

    dev = _dev()
    gen = torch.Generator().manual_seed(7 * b + K + EIN)
    R, H = b * K, 32
    dH0 = torch.randn(R, H, generator=gen).to(dev)
    dSocR = torch.randn(R, max(S, 1), generator=gen).to(dev)
    W = (torch.randn(H, EIN + Z, generator=gen) * 0.3).to(dev)
    inv = torch.stack([torch.randperm(R, generator=gen)[:b] for _ in range(K)]).reshape(-1).to(torch.int32).to(dev)  # (K*b)
    st = torch.cuda.current_stream().cuda_stream
    p = HF._p
    dQe0, dEnc0 = torch.empty(b, H, device=dev), torch.empty(b, EIN, device=dev)
    lib.mggan_gather_sum(p(dH0), H, p(inv), p(dQe0), H, b, K, H, 0, st)
    lib.mggan_linear_bwd_data(p(dQe0), H, p(W), EIN + Z, p(dEnc0), EIN, b, EIN, H, 0, 0, 0, HF.ACT_NONE, 0.0, st)
    if S:
        lib.mggan_gather_sum(p(dSocR), S, p(inv), dEnc0.data_ptr() + 4 * (EIN - S), EIN, b, K, S, 1, st)
    dQe1, dEnc1 = torch.full((b, H), float("nan"), device=dev), torch.full((b, EIN), float("nan"), device=dev)
    lib.mggan_rollout_ped_adjoint(p(dH0), p(dSocR) if S else None, p(inv), p(W), EIN + Z, p(dQe1), p(dEnc1), EIN, b, K, EIN, S, st)
    torch.cuda.synchronize()
    assert torch.equal(dQe1, dQe0)  # the same four-way sums in the same order
    torch.testing.assert_close(dEnc1, dEnc0, rtol=1e-5, atol=1e-5)


def _patch_gram_f64(img):
    """P[s][t] = sum over images and positions of patch[s] * patch[t]; tap t = 9*ci + 3*ky + kx, tap 36 = 1 (f64, torch)."""
    x = img.double()
    pat = torch.nn.functional.unfold(x, kernel_size=3, padding=1)  # (B, 36, 1089), rows ordered (ci, ky, kx)
    pat = torch.cat([pat, torch.ones(x.shape[0], 1, pat.shape[2], dtype=torch.float64)], 1)
    return torch.einsum("bsp,btp->st", pat, pat)


@pytest.mark.parametrize("B,real", [(1, None), (3, None), (50, None), (777, None), (1500, None), (40, 29), (16, 0)])
def test_image_gram_matches_the_patch_product(B, real):
    """Both Gram kernels (autocorrelation form = the default, MFMA tap-by-tap = MGGAN_GRAM_KERNEL=mfma, exercised by
    tools/ab_gram.py) against the f64 patch product; a padded batch counts its real images only."""
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(B)
    img = torch.randn(B, 4, 33, 33, generator=g) * 0.7 + 0.2
    img[:, 3] = (img[:, 3] > 0.5).float()  # a sparse plane, like the rasterised pedestrians
    imgd = img.to(dev)
    gram = torch.full((37 * 37,), float("nan"), dtype=torch.float64, device=dev)
    nb = lib.mggan_image_gram_workspace(B)
    ws = torch.empty(nb // 8, dtype=torch.float64, device=dev)
    dims = 0
    if real is not None:
        dimst = torch.tensor([real, 1, 0, 0], dtype=torch.int32, device=dev)
        dims = dimst.data_ptr()
    lib.mggan_image_gram(imgd.data_ptr(), B, gram.data_ptr(), ws.data_ptr(), nb, dims, st())
    ref = _patch_gram_f64(img[:B if real is None else real]) if (real is None or real > 0) else torch.zeros(37, 37, dtype=torch.float64)
    got = gram.cpu().view(37, 37)
    assert torch.isfinite(got).all()
    scale = ref.abs().max().clamp_min(1.0)
    assert (got - ref).abs().max() / scale < 2e-6
    assert torch.equal(got, got.t())


@pytest.mark.parametrize("C,B,real", [(16, 1, None), (16, 37, None), (8, 37, None), (8, 300, None), (16, 300, 211), (8, 5, 0)])
def test_conv1_wgrad_rows_match_the_sparse_patch_product(C, B, real):
    """The sparse part of the conv1 weight gradient, A[c][t] = sum over images and pooled cells of g[c][cell] * patch[position the
    cell's code selects][t] (mggan_conv1_wgrad with dW = NULL: one f64 row per workgroup; the default kernel gathers on the vector
    ALU, MGGAN_C1WGRAD=mfma is the one-hot matrix form -- tools/ab_c1wgrad.py runs both), against torch f64."""
    lib, dev = _lib(), _dev()
    g = torch.Generator().manual_seed(100 * C + B)
    img = torch.randn(B, 4, 33, 33, generator=g) * 0.7 + 0.2
    G1c = torch.randn(B, C, 16, 16, generator=g)
    code = torch.randint(0, 4, (B, C, 16, 16), generator=g, dtype=torch.uint8)
    rows = max(lib.mggan_cnn_grid(B), 1)
    ws = torch.full((rows * C * 36,), float("nan"), dtype=torch.float64, device=dev)
    imgd, Gd, cd = img.to(dev), G1c.to(dev), code.to(dev)
    dims = 0
    if real is not None:
        dimst = torch.tensor([real, 1, 0, 0], dtype=torch.int32, device=dev)
        dims = dimst.data_ptr()
    lib.mggan_conv1_wgrad(imgd.data_ptr(), B, C, Gd.data_ptr(), cd.data_ptr(), 0, 0, 0, 0, 0, ws.data_ptr(), ws.numel() * 8, 0,
                          dims, st())
    got = ws.view(rows, C * 36).sum(0).view(C, 36).cpu()
    n = B if real is None else real
    ref = torch.zeros(C, 36, dtype=torch.float64)
    if n > 0:
        dy = torch.zeros(n, C, 34, 34, dtype=torch.float64)
        c64 = code[:n].long()
        yy = 2 * torch.arange(16).view(1, 1, 16, 1) + (c64 >> 1)
        xx = 2 * torch.arange(16).view(1, 1, 1, 16) + (c64 & 1)
        dy.view(n, C, -1).scatter_(2, (yy * 34 + xx).view(n, C, -1), G1c[:n].double().view(n, C, -1))
        pat = torch.nn.functional.unfold(img[:n].double(), kernel_size=3, padding=1).view(n, 36, 33, 33)
        ref = torch.einsum("bcyx,btyx->ct", dy[:, :, :33, :33], pat)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2e-6 * ref.abs().max().clamp_min(1.0)
