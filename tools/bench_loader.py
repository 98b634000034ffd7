#!/usr/bin/env python
"""Input-pipeline throughput: pedestrians per second through get_dataloader on a synthetic on-disk
dataset in the reference's format (one 640x480 scene, 120 frames x 40 pedestrians), host crops (PIL, as the
reference) against device crops (scene images resident in HBM, csrc/crop.hip).
    python tools/bench_loader.py [--scenes-per-batch 64]"""
import argparse
import io
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-gan_amd"))


def make_dataset(root, frames=120, peds=40, w=640, h=480, phase="test"):
    r = np.random.RandomState(0)
    d = Path(root) / "eth" / phase
    d.mkdir(parents=True)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.clip(np.stack([127 + 100 * np.sin(xx / 17.0), 127 + 100 * np.cos(yy / 11.0), (xx * 3 + yy * 5) % 256], -1) +
                  r.randn(h, w, 3) * 10, 0, 255).astype(np.uint8)
    Image.fromarray(img).save(d / "plaza.jpg", quality=92)
    p = np.stack([r.uniform(3, w * 0.05 - 3, peds), r.uniform(3, h * 0.05 - 3, peds)], 1)
    v = r.randn(peds, 2) * 0.05
    with open(d / "biwi_plaza.txt", "w") as fh:
        for f in range(frames):
            p = p + v + r.randn(peds, 2) * 0.01
            for i in range(peds):
                fh.write("{:.1f}\t{:.1f}\t{:.6f}\t{:.6f}\n".format(f, i + 1, p[i, 1], p[i, 0]))


def run(loader, n_batches, sync):
    it = iter(loader)
    next(it)  # warm-up
    t0, peds = time.perf_counter(), 0
    for _ in range(n_batches):
        b = next(it)
        peds += b["features"].shape[0]
    if sync:
        torch.cuda.synchronize()
    return peds / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes-per-batch", type=int, default=32)
    ap.add_argument("--batches", type=int, default=2)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="mggan_loader_")
    os.environ["MGGAN_DATA_ROOT"] = tmp
    make_dataset(tmp)
    from mggan.data_utils.data_loaders import get_dataloader

    host = run(get_dataloader("eth", "test", batch_size=a.scenes_per_batch), a.batches, False)
    print("host crops   (PIL per pedestrian, 2 image resizes per scene): {:9.0f} pedestrians/s".format(host))
    if torch.cuda.is_available():
        dev = run(get_dataloader("eth", "test", batch_size=a.scenes_per_batch, crop_device="cuda"), a.batches, True)
        print("device crops (scene images in HBM, one launch per batch):     {:9.0f} pedestrians/s  ({:.1f}x)".format(dev, dev / host))
    # training with --augment 1 (the reference's default): flip + rotate(expand) + Lanczos resize of the scene image per item
    make_dataset(tmp, phase="train")
    np.random.seed(0)
    host_a = run(get_dataloader("eth", "train", augment=True, batch_size=a.scenes_per_batch), a.batches, False)
    print("host crops, augmented   (Pillow rotate + 2 resizes per item):   {:9.0f} pedestrians/s".format(host_a))
    if torch.cuda.is_available():
        np.random.seed(0)
        ld = get_dataloader("eth", "train", augment=True, batch_size=a.scenes_per_batch, crop_device="cuda")
        run(ld, a.batches, True)  # (first pass: the Lanczos tables of the sizes it meets are built and cached)
        dev_a = run(ld, max(a.batches, 2), True)
        print("device crops, augmented (per-crop taps from the resident image): {:9.0f} pedestrians/s  ({:.1f}x)".format(
            dev_a, dev_a / host_a))


if __name__ == "__main__":
    main()
