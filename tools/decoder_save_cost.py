"""The rollout forward at 163,840 rows with and without its saved state (no_grad vs grad), timed alone with HIP events;
MGGAN_DEC_FWD=4 / 1 selects the kernel."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "mg-gan_amd")]
import torch
import bench
from mggan.data_utils import synthetic
from mggan.hip.lib import start_trace, stop_trace

dev = torch.device("cuda", 0)
tr = bench.build_trainer(8, "device", dev)
batch = tr.to_device(synthetic.make_batch(synthetic.scene_sizes(256, 32), seed=0))
sse = batch["seq_start_end"]
tr.rng.plan = (1, 20, 1)
for mode in ("nograd", "grad", "nograd", "grad"):
    tr.rng.begin_iteration(sse, 8192, 8, dev)
    start_trace()
    ctx = torch.no_grad() if mode == "nograd" else torch.enable_grad()
    with ctx:
        out = tr.G(batch["in_xy"], batch["in_dxdy"], sse, noise=None, all_gen_out=False, img=batch["features"], num_samples=20)
    t = stop_trace()
    c, ms, a, _ = t["mggan_decoder_rollout_fwd"]
    print(mode, "decoder_rollout_fwd ms:", [round(x, 4) for x in ms], "rows", a[0][0], flush=True)
    del out
