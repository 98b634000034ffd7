"""conv1 + pooling decision (mggan_conv1_pool) alone on this box: average time of 50 launches at 1,536 and 8,192 images (no fused
finalize: ticket = NULL), its statistics and selected values against torch f64.  AB_C=16 times the generator's width; with
MGGAN_HIP_LIB pointing at another build of the library (tools/build_rev.sh) it is the A/B harness round 5 used for the
vector-ALU form of the C = 8 kernel (built, 166 us against the matrix form's 128 us, not kept)."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd"))
from mggan.hip import lib  # noqa: E402

dev = torch.device("cuda")
C = int(os.environ.get("AB_C", "8"))
for B in (1536, 8192):
    g = torch.Generator().manual_seed(B)
    img = (torch.randn(B, 4, 33, 33, generator=g) * 0.7 + 0.2).to(dev)
    W = (torch.randn(C, 4, 3, 3, generator=g) * 0.2).to(dev)
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    gamma = torch.randn(C, generator=g).to(dev)
    xsel = torch.empty(B, C, 16, 16, device=dev)
    code = torch.empty(B, C, 16, 16, dtype=torch.uint8, device=dev)
    part = torch.zeros(lib.mggan_cnn_grid(B) * 2 * C + 64, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.mggan_conv1_pool(img.data_ptr(), B, C, W.data_ptr(), bias.data_ptr(), xsel.data_ptr(), code.data_ptr(),
                                       part.data_ptr(), 0, float(B) * 1089, gamma.data_ptr(), 0, 0, 0, 0, 0.1, 1e-5, 1, 0, 0, 0,
                                       0, 0, 0, s)
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    rows = part[:lib.mggan_cnn_grid(B) * 2 * C].view(-1, 2 * C).sum(0)
    y = torch.nn.functional.conv2d(img.double(), W.double(), bias.double(), padding=1)
    ref = torch.cat([y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))])
    sg = torch.where(gamma < 0, -1.0, 1.0).double().view(1, C, 1, 1)
    pooled = sg * torch.nn.functional.max_pool2d(sg * y[:, :, :32, :32], 2)
    print(f"lib={os.path.basename(os.environ.get('MGGAN_HIP_LIB', 'in-tree'))} C={C} B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call; "
          f"statistics rel dev {((rows - ref).abs() / ref.abs()).max().item():.2e}, selected values max dev "
          f"{(xsel.double() - pooled).abs().max().item():.2e}")
