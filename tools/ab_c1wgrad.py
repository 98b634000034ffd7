"""conv1 weight gradient, sparse part (mggan_conv1_wgrad with dW = NULL: the per-workgroup rows A[c][t]) alone on this box:
run once per MGGAN_C1WGRAD value (gather = default | mfma; read once per process), AB_C=8|16.  Prints the average time of 50
launches at 1,536 and 8,192 images and the deviation of the folded rows from an f64 torch reference."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd"))
from mggan.hip import lib  # noqa: E402

dev = torch.device("cuda")
C = int(os.environ.get("AB_C", "16"))
for B in (1536, 8192):
    g = torch.Generator().manual_seed(B + C)
    img = (torch.randn(B, 4, 33, 33, generator=g) * 0.7 + 0.2).to(dev)
    G1c = torch.randn(B, C, 16, 16, generator=g).to(dev)
    code = torch.randint(0, 4, (B, C, 16, 16), generator=g, dtype=torch.uint8).to(dev)
    rows = lib.mggan_cnn_grid(B)
    ws = torch.zeros(rows * C * 36, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.mggan_conv1_wgrad(img.data_ptr(), B, C, G1c.data_ptr(), code.data_ptr(), 0, 0, 0, 0, 0, ws.data_ptr(),
                                        ws.numel() * 8, 0, 0, s)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    got = ws.view(rows, C * 36).sum(0).view(C, 36)
    ref = torch.zeros(C, 36, dtype=torch.float64, device=dev)
    for lo in range(0, B, 256):
        n = min(256, B - lo)
        dy = torch.zeros(n, C, 34, 34, dtype=torch.float64, device=dev)
        cd = code[lo:lo + n].long()
        yy = 2 * torch.arange(16, device=dev).view(1, 1, 16, 1) + (cd >> 1)
        xx = 2 * torch.arange(16, device=dev).view(1, 1, 1, 16) + (cd & 1)
        dy.view(n, C, -1).scatter_(2, (yy * 34 + xx).view(n, C, -1), G1c[lo:lo + n].double().view(n, C, -1))
        pat = torch.nn.functional.unfold(img[lo:lo + n].double(), kernel_size=3, padding=1).view(n, 36, 33, 33)
        ref += torch.einsum("bcyx,btyx->ct", dy[:, :, :33, :33], pat)
    print(f"MGGAN_C1WGRAD={os.environ.get('MGGAN_C1WGRAD', '(default: gather)')} C={C} B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us "
          f"per call; max deviation {((got - ref).abs().max() / ref.abs().max()).item():.2e} of the largest entry")
