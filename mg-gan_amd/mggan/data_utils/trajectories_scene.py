"""On-disk trajectory datasets -> collated scene batches.

Surface of the reference's loader (/root/reference/mggan/data_utils/trajectories_scene.py:40-371 and
BaseTrajectories.py:23-288): `TrajectoryDatasetEval(dataset_name=..., phase=..., margin_in=16, margin_out=16,
scaling_small=..., data_augmentation=0|1)` with `.trajectory (N,20,2)`, `.seq_start_end`, `.scene_list`, `.ped_ids`,
`.images`, `__getitem__` -> [obs, pred, obs_rel, pred_rel, scene images, features (n,4,33,33), walls] and the
collate function `seq_collate_scene`.  Host work (pandas / numpy / PIL) by nature; the per-pedestrian image crops --
the part that scales with the batch -- can instead be cut on the GPU from scene images kept in HBM
(`mggan.data_utils.device_crops`, csrc/crop.hip).

A sequence = 20 consecutive frames of one text file (sliding window, stride `skip`); a pedestrian belongs to it
only when present in all 20 frames; pedestrians flagged inactive (GOFP) keep their history but get NaN ground
truth."""
import math
import os
from collections import defaultdict

import numpy as np
import pandas as pd
import torch
from PIL import Image
from torch.utils.data import Dataset

from mggan.data_utils import experiments

RESAMPLE = getattr(Image, "Resampling", Image).LANCZOS  # what Pillow < 10 called Image.ANTIALIAS


def rotate(points, center, alpha):
    """Rotate (n,2) points by alpha (radians, image convention: y down) around `center`."""
    c, s = np.cos(alpha), np.sin(alpha)
    d = points - np.asarray(center)[None]
    out = points.copy()
    out[:, 0] = d[:, 0] * c + d[:, 1] * s + center[0]
    out[:, 1] = -d[:, 0] * s + d[:, 1] * c + center[1]
    return out


def _flat(items):
    for it in items:
        if isinstance(it, (list, tuple)):
            yield from _flat(it)
        else:
            yield it


def seq_collate_scene(data):
    """List of dataset items -> the batch dictionary of trajectories_scene.py:40-78 (time-major tensors,
    `seq_start_end` as a python list of [start, end])."""
    obs, pred, obs_rel, pred_rel, scene_imgs, feats, occupancy = zip(*data)
    bounds = np.cumsum([0] + [len(o) for o in obs]).tolist()
    tm = lambda parts: torch.cat(parts, dim=0).permute(1, 0, 2)
    try:
        features = torch.cat(feats, dim=0)
    except (RuntimeError, TypeError):
        features = torch.empty(1)
    in_xy = tm(obs)
    return {"in_xy": in_xy, "gt_xy": tm(pred), "in_dxdy": tm(obs_rel), "gt_dxdy": tm(pred_rel),
            "size": torch.LongTensor([in_xy.size(1)]), "scene_img": tuple(_flat(scene_imgs)), "features": features,
            "occupancy": occupancy, "seq_start_end": [[s, e] for s, e in zip(bounds, bounds[1:])]}


class BaseDataset(Dataset):
    def __init__(self, save=False, load_p=True, dataset_name="stanford", phase="test", obs_len=8, pred_len=12,
                 time_step=0.4, skip=1, data_augmentation=0, scale_img=True, max_num=None, load_occupancy=False,
                 logger=None, special_scene=None, scaling_small=0.5, scaling_tiny=0.25, margin_in=32, margin_out=16,
                 margin_tiny=8, **kwargs):
        super().__init__()
        self.dataset_name, self.phase, self.obs_len, self.pred_len = dataset_name, phase, obs_len, pred_len
        self.time_step, self.skip, self.data_augmentation = time_step, skip, data_augmentation
        self.load_occupancy, self.special_scene = load_occupancy, special_scene
        self.scaling_small, self.scaling_tiny = scaling_small, scaling_tiny
        self.margin_in, self.margin_out, self.margin_tiny = margin_in, margin_out, margin_tiny
        self.dataset = experiments.get(dataset_name)
        self.__dict__.update(self.dataset.get_dataset_args())
        self.data_dir = self.dataset.get_file_path(phase)
        self.seq_len = obs_len + pred_len
        self.images = {}
        self.all_files = [os.path.join(self.data_dir, f) for f in os.listdir(self.data_dir)]

    # ---- scene images (BaseTrajectories.py:67-125) ---------------------------------------------------
    def load_image(self, path, scene):
        img = Image.open(path)
        scaled, factor, ratio = img, 1, 1.0
        if "stanford" in self.dataset_name or "gofp" in self.dataset_name:
            if "stanford" in self.dataset_name:
                h = self.homography
                ratio = h.loc[(h["File"] == "{}.jpg".format(scene)) & (h["Version"] == "A"), "Ratio"].iloc[0]
            else:
                ratio = self.homography[scene]
            factor = ratio / self.img_scaling  # metres per pixel -> img_scaling metres per pixel
            scaled = img.resize((int(round(img.size[0] * factor)), int(round(img.size[1] * factor))), RESAMPLE)
        self.images[scene] = {"ratio": ratio, "scale_factor": factor, "scaled_image": scaled,
                              "small_image": self._resized(scaled, self.scaling_small),
                              "tiny_image": self._resized(scaled, self.scaling_tiny)}

    def _resized(self, img, metres_per_pixel):
        f = self.img_scaling / metres_per_pixel
        return img.resize((int(round(img.width * f)), int(round(img.height * f))), RESAMPLE)

    def get_ratio(self, scene):
        return self.images[scene]["ratio"]

    def scale2meters(self):
        self.trajectory *= self.img_scaling
        self.format = "meter"

    # ---- text files (BaseTrajectories.py:134-161) ----------------------------------------------------
    def load_file(self, path, delim="tab"):
        """-> float array with columns frame, ID, x, y (+ is_active)."""
        df = pd.read_csv(path, header=None, delimiter={"tab": "\t", "space": " "}.get(delim, delim))
        df.columns = self.data_columns
        if "lost" in df:  # SDD annotations: pedestrians that are not lost
            df = df[(df["label"] == "Pedestrian") & (df["lost"] == 0)]
        if self.dataset_name in ("stanford", "gofp"):  # annotation rate -> one frame per time step
            every = int(round(self.framerate * self.time_step))
            df = df[df["frame"] % every == 0].copy()
            df["frame"] /= every
        cols = ["frame", "ID", "x", "y"] + (["is_active"] if "is_active" in self.data_columns else [])
        return np.asarray(df[cols].values)

    def __len__(self):
        return len(self.seq_start_end)

    # ---- per-pedestrian crop of the scene image (BaseTrajectories.py:254-288) -------------------------
    def crop_center(self, last_obs):
        """Pixel (x, y) of the crop centre in the small image for a last observed position."""
        scale = 1.0 / self.scaling_small if self.format == "meter" else 1
        return (np.asarray(last_obs, dtype=np.float32) * scale).astype(int)

    def ImageFeatures_small(self, scene_image, trajectory, prediction, image_type="small_image"):
        """-> ((1,4,2m+1,2m+1) RGB in [-1,1) + one-hot centre channel, the PIL crop)."""
        m = self.margin_in
        xc, yc = self.crop_center(trajectory[-1].cpu().numpy())
        crop = scene_image[image_type].crop((int(xc - m), int(yc - m), int(xc + m + 1), int(yc + m + 1)))
        rgb = -1 + torch.from_numpy(np.array(crop) * 1.0) * 2.0 / 256
        centre = torch.zeros(2 * self.margin_out + 1, 2 * self.margin_out + 1, 1)
        centre[m, m, 0] = 1
        return torch.cat((rgb.float(), centre), dim=2).permute(2, 0, 1).unsqueeze(0), crop


class TrajectoryDatasetEval(BaseDataset):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.scene_list, self.image_list, self.wall_points_dict, self.walls_list = [], [], {}, []
        for path in (f for f in self.all_files if ".jpg" in f):
            scene = os.path.basename(path).split(".")[0]
            is_occupancy = scene.split("-")[-1] == "op"
            if self.load_occupancy and is_occupancy:
                self.load_image(path, scene.split("-")[-2])
            elif not self.load_occupancy and not is_occupancy:
                self.load_image(path, scene)
        assert self.images, "No valid imges in folder"

        ped_ids, seqs, counts = [], [], []
        for path in (f for f in self.all_files if ".txt" in f):
            if self.special_scene and self.special_scene not in path:
                continue
            scene = "_".join(os.path.basename(path).split(".")[0].split("_")[1:])
            rows = self.load_file(path, self.delim)
            by_frame = defaultdict(list)
            for r in rows:
                by_frame[r[0]].append(r)
            frames = [np.stack(by_frame[f]) for f in sorted(by_frame)]
            n_windows = int(math.ceil((len(frames) - self.seq_len) / self.skip))
            for start in range(0, n_windows * self.skip, self.skip):
                window = np.concatenate(frames[start:start + self.seq_len], axis=0)
                peds = []
                for pid in np.unique(window[:, 1]):
                    track = window[window[:, 1] == pid]
                    if len(track) != self.seq_len or (np.diff(track[:, 0]) != 1).any():
                        continue  # not present in every frame of the window
                    xy = track[:, 2:4].copy()
                    if track.shape[1] == 5 and (track[:, 4] == 0).any():
                        xy[self.obs_len:] = np.nan  # inactive: observed but never evaluated
                    ped_ids.append(pid)
                    peds.append(xy)
                if peds:
                    counts.append(len(peds))
                    seqs.append(np.stack(peds, axis=0))
                    self.scene_list.append(scene)
        self.ped_ids = np.array(ped_ids, int)
        bounds = np.cumsum([0] + counts).tolist()
        self.seq_start_end = list(zip(bounds, bounds[1:]))
        self.trajectory = np.concatenate(seqs, axis=0)
        print("scene list", len(self.scene_list))
        print("trajectories", len(self.trajectory))
        if self.scale:  # pixel datasets: coordinates follow the rescaled image
            for (s, e), scene in zip(self.seq_start_end, self.scene_list):
                self.trajectory[s:e] *= self.images[scene]["scale_factor"]
        if self.norm2meters:
            print("norming to meters")
            self.scale2meters()
        self.wall_available = False

    def _rel(self):
        return self.trajectory[:, 1:] - self.trajectory[:, :-1]

    @property
    def obs_traj(self):
        return torch.from_numpy(self.trajectory[:, :self.obs_len]).float()

    @property
    def pred_traj(self):
        return torch.from_numpy(self.trajectory[:, self.obs_len:]).float()

    @property
    def obs_traj_rel(self):
        return torch.from_numpy(self._rel()[:, :self.obs_len - 1]).float()

    @property
    def pred_traj_rel(self):
        return torch.from_numpy(self._rel()[:, self.obs_len - 1:]).float()

    def get_scene(self, index):
        in_xy, gt_xy, in_dxdy, gt_dxdy, scene_img, features, _ = self[index]
        tm = lambda t: t.permute(1, 0, 2)
        return {"in_xy": tm(in_xy), "gt_xy": tm(gt_xy), "in_dxdy": tm(in_dxdy), "gt_dxdy": tm(gt_dxdy),
                "scene_img": scene_img, "features": features.squeeze(0), "seq_start_end": [[0, in_xy.size(0)]]}

    def augmentation(self):
        """(rotation angle, flip code 0/1/2) -- two draws from numpy's global generator per training item."""
        if self.data_augmentation and self.phase == "train":
            return np.random.rand() * 2 * np.pi, int(np.random.choice(np.arange(3)))
        return 0, 0

    def transformed_xy(self, index, alpha, flip, size):
        """Coordinates of item `index` in the frame of its scene image (scaled size `size` = (w, h)) after the
        flip / rotation of trajectories_scene.py:272-312; returns (xy (n,20,2) float64, canvas offset)."""
        start, end = self.seq_start_end[index]
        scene = self.scene_list[index]
        xy = self.trajectory[start:end].copy()
        if self.format == "pixel":
            to_orig = 1 / self.images[scene]["scale_factor"]
        elif self.format == "meter":
            to_orig = self.img_scaling
        else:
            raise AssertionError(" Not valid format '{}': 'meters' or 'pixel'".format(self.format))
        w, h = size
        center = np.array([w, h]) / 2.0
        corners = np.array([[0, 0], [0, h], [w, h], [w, 0]])
        if flip == 1:
            xy[:, :, 0] = w * to_orig - xy[:, :, 0]
        elif flip == 2:
            xy[:, :, 1] = h * to_orig - xy[:, :, 1]
        offset = rotate(corners, center, alpha).min(axis=0)  # the expanded canvas starts at the rotated corners' minimum
        flat = xy.reshape((end - start) * self.seq_len, -1)
        return (rotate(flat.copy(), center * to_orig, alpha) - offset * to_orig).reshape(end - start, self.seq_len, -1)

    def transformed(self, index, alpha, flip):
        """Scene image and coordinates of item `index` after flip / rotation."""
        scene = self.scene_list[index]
        img = self.images[scene]["scaled_image"]
        xy = self.transformed_xy(index, alpha, flip, img.size)
        if flip == 1:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        elif flip == 2:
            img = img.transpose(Image.FLIP_TOP_BOTTOM)
        return scene, img.rotate(alpha / np.pi * 180, expand=True), xy

    def __getitem__(self, index):
        start, end = self.seq_start_end[index]
        alpha, flip = self.augmentation()
        scene, img, xy = self.transformed(index, alpha, flip)
        scene_image = {"ratio": self.images[scene]["ratio"], "scene": scene, "scaled_image": img.copy(),
                       "small_image": self._resized(img, self.scaling_small),
                       "tiny_image": self._resized(img, self.scaling_tiny)}
        xy = torch.from_numpy(xy).float()
        dxdy = xy[:, 1:] - xy[:, :-1]
        obs, pred = xy[:, :self.obs_len], xy[:, self.obs_len:]
        features = torch.cat([self.ImageFeatures_small(scene_image, obs[i], pred[i])[0] for i in range(end - start)])
        return [obs, pred, dxdy[:, :self.obs_len - 1], dxdy[:, self.obs_len - 1:], (end - start) * [scene_image], features,
                torch.empty(1)]
