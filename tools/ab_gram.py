"""Gram kernels side by side on this box: `python tools/ab_gram.py` once per MGGAN_GRAM_KERNEL value (the choice is read
once per process).  Prints max relative deviation from the f64 patch product and the average time of 50 launches."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd"))
from mggan.hip import lib  # noqa: E402

dev = torch.device("cuda")
for B in (1536, 8192):
    g = torch.Generator().manual_seed(B)
    img = (torch.randn(B, 4, 33, 33, generator=g) * 0.7 + 0.2).to(dev)
    gram = torch.empty(37 * 37, dtype=torch.float64, device=dev)
    nb = lib.mggan_image_gram_workspace(B)
    ws = torch.empty(nb // 8, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.mggan_image_gram(img.data_ptr(), B, gram.data_ptr(), ws.data_ptr(), nb, 0, s)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    x = img.double()
    ref = torch.zeros(37, 37, dtype=torch.float64, device=dev)
    for lo in range(0, B, 512):
        pat = torch.nn.functional.unfold(x[lo:lo + 512], kernel_size=3, padding=1)
        pat = torch.cat([pat, torch.ones(pat.shape[0], 1, pat.shape[2], dtype=torch.float64, device=dev)], 1)
        ref += torch.einsum("bsp,btp->st", pat, pat)
    err = ((gram.view(37, 37) - ref).abs().max() / ref.abs().max()).item()
    print(f"MGGAN_GRAM_KERNEL={os.environ.get('MGGAN_GRAM_KERNEL', '(default)')} B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, "
          f"max rel deviation {err:.2e}")
