"""Golden vectors for the on-disk dataset loader: small synthetic datasets in the reference's
on-disk format are written to a temporary directory and read by the REAL reference loader
(/root/reference/mggan/data_utils/trajectories_scene.py TrajectoryDatasetEval + seq_collate_scene); the fixture
stores the dataset FILES (so the tests can rebuild the directory anywhere) and what the reference made of them.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_loader.py       -> tests/golden/golden_loader.npz"""
import io
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

_refload.install_stubs()
np.int = int                       # aliases the reference still uses (trajectories_scene.py:184)
Image.ANTIALIAS = Image.LANCZOS    # (BaseTrajectories.py:92,104,109); same filter under its current name
import mggan.data_utils.experiments as ref_exp  # noqa: E402
from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval, seq_collate_scene  # noqa: E402


def scene_image(w, h, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 17.0 + seed), 127 + 100 * np.cos(yy / 11.0), (xx * 3 + yy * 5) % 256], -1)
    img = np.clip(img + r.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=92)
    return buf.getvalue()


def tracks(n_frames, n_peds, w, h, seed, frame_step=1):
    """rows (frame, id, x, y, active) of pedestrians that enter and leave at different times."""
    r = np.random.RandomState(seed)
    rows = []
    for pid in range(n_peds):
        f0 = int(r.randint(0, 6)) if pid % 3 else 0
        f1 = n_frames - (int(r.randint(0, 6)) if pid % 4 == 1 else 0)
        p = np.array([r.uniform(0.2, 0.8) * w, r.uniform(0.2, 0.8) * h])
        v = r.randn(2) * 0.012 * w
        inactive_from = f0 + 14 if pid == 2 else None
        for f in range(f0, f1):
            p = p + v + r.randn(2) * 0.002 * w
            rows.append((f * frame_step, pid + 1, p[0], p[1], 0 if inactive_from is not None and f >= inactive_from else 1))
    rows.sort(key=lambda t: (t[0], t[1]))
    return rows


def build(root):
    files = {}
    # BiWi format (eth): metres, columns frame, ID, y, x
    rows = tracks(30, 6, 24.0, 18.0, seed=1)
    files["eth/{p}/biwi_sceneA.txt"] = "".join("{:.1f}\t{:.1f}\t{:.6f}\t{:.6f}\n".format(f, i, y, x) for f, i, x, y, _ in rows)
    files["eth/{p}/sceneA.jpg"] = scene_image(480, 360, 3)
    files["eth/{p}/sceneB.jpg"] = scene_image(320, 240, 4)
    # GOFP format: pixels at 10 fps, columns frame, ID, x, y, moment, old frame, old_ID, is_active
    rows = tracks(30, 5, 400.0, 300.0, seed=2, frame_step=4)
    files["gofp/{p}/x_zara1.txt"] = "".join("{}\t{}\t{:.4f}\t{:.4f}\t0\t{}\t{}\t{}\n".format(f, i, x, y, f, i, a)
                                            for f, i, x, y, a in rows)
    files["gofp/{p}/zara1.jpg"] = scene_image(400, 300, 5)
    # SDD format: pixels at 30 fps, 12 columns, pedestrians that are not lost
    rows = tracks(28, 5, 500.0, 380.0, seed=3, frame_step=12)
    lines = []
    for f, i, x, y, _ in rows:
        label = "Biker" if i == 4 else "Pedestrian"
        lines.append("{}\t0\t0\t0\t0\t{}\t{}\t0\t0\t{}\t{:.4f}\t{:.4f}\n".format(i, f, 1 if (i == 5 and f > 200) else 0, label, x, y))
        if f % 12 == 0:
            lines.append("{}\t0\t0\t0\t0\t{}\t0\t0\t0\tPedestrian\t{:.4f}\t{:.4f}\n".format(i, f + 5, x + 1, y + 1))  # off-step frame
    files["stanford/{p}/video0_quad.txt"] = "".join(lines)
    files["stanford/{p}/quad.jpg"] = scene_image(500, 380, 6)
    files["stanford/H_SDD.txt"] = "File\tVersion\tRatio\nquad.jpg\tA\t0.0375\nquad.jpg\tB\t0.05\n"
    out = {}
    for rel, data in files.items():
        for phase in (("train", "test") if "{p}" in rel else ("",)):
            path = Path(root) / rel.format(p=phase)
            path.parent.mkdir(parents=True, exist_ok=True)
            mode = "wb" if isinstance(data, bytes) else "w"
            with open(path, mode) as fh:
                fh.write(data)
            out["file/" + rel.format(p=phase)] = np.frombuffer(data if isinstance(data, bytes) else data.encode(), dtype=np.uint8)
    return out


def t2n(t):
    return t.detach().cpu().numpy().copy()


def main():
    tmp = tempfile.mkdtemp(prefix="mggan_ds_")
    ref_exp.root_path = Path(tmp)            # Experiment.__init__ resolves <root>/data/datasets/<name>
    out = build(Path(tmp) / "data" / "datasets")
    for name, small in (("eth", 0.5), ("gofp", 0.5), ("stanford", 0.7)):   # data_loaders.py:60-95
        for phase, aug in (("test", 0), ("train", 1)):
            ds = TrajectoryDatasetEval(dataset_name=name, phase=phase, margin_in=16, margin_out=16, load_occupancy=False,
                                       scaling_small=small, data_augmentation=aug)
            p = "{}/{}/".format(name, phase)
            out[p + "trajectory"] = ds.trajectory.copy()
            out[p + "seq_start_end"] = np.array(ds.seq_start_end, dtype=np.int64)
            out[p + "ped_ids"] = np.asarray(ds.ped_ids, dtype=np.int64)
            out[p + "scenes"] = np.array(ds.scene_list)
            np.random.seed(123)
            batch = seq_collate_scene([ds[i] for i in range(min(3, len(ds)))])
            for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy", "features"):
                out[p + "batch/" + k] = t2n(batch[k])
            out[p + "batch/seq_start_end"] = np.array(batch["seq_start_end"], dtype=np.int64)
            print(name, phase, "sequences", len(ds), "peds", len(ds.trajectory), "features", tuple(batch["features"].shape),
                  "nan peds", int(np.isnan(ds.trajectory).any((1, 2)).sum()))
    path = os.path.join(HERE, "golden_loader.npz")
    np.savez_compressed(path, **out)
    print("bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
