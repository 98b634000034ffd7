"""Dataset descriptions: where the files of a dataset live and how its text rows are laid out
(surface of /root/reference/mggan/data_utils/experiments.py:28-507, only what the loader reads).

On-disk format: `<root>/<name>/{train,val,test}/` holds tab-separated `*.txt` trajectory files and one
`<scene>.jpg` per scene; a text file `<anything>_<scene>.txt` belongs to the image `<scene>.jpg`
(trajectories_scene.py:107-135).  `<root>` is `$MGGAN_DATA_ROOT`, default `<repo>/data/datasets` like the
reference's `data/datasets`."""
import os
from pathlib import Path

import pandas as pd

BIWI_COLUMNS = ["frame", "ID", "y", "x"]                                   # experiments.py:185
SDD_COLUMNS = ["ID", "xmin, left", "ymin, left", "xmax, right", "ymax, right", "frame", "lost", "occuluded", "generated",
               "label", "x", "y"]                                          # experiments.py:200-213
GOFP_COLUMNS = ["frame", "ID", "x", "y", "moment", "old frame", "old_ID", "is_active"]  # experiments.py:481-490
GOFP_RATIOS = {"zara1": 0.03109532180986424, "eth": 0.06668566952360758, "hotel": 0.0225936169079401,
               "0000": 0.042200689823829046, "0400": 0.07905284109247492, "0401": 0.0598454105469989,
               "0500": 0.04631904070838066, "zara2": 0.03109532180986424}  # metres per pixel, experiments.py:496-504


def data_root():
    return Path(os.environ.get("MGGAN_DATA_ROOT", Path(__file__).resolve().parents[3] / "data" / "datasets"))


class Experiment:
    """name -> directory + the arguments the dataset class needs (`get_dataset_args`)."""

    columns, fmt, norm2meters, scale, framerate = BIWI_COLUMNS, "meter", False, False, None

    def __init__(self):
        self.name = type(self).__name__
        self.data_path = data_root() / self.name

    def get_file_path(self, phase):
        if phase not in ("train", "val", "test"):
            raise AssertionError('"phase" must be either train, val or test.')
        return str(self.data_path / phase)

    def homography(self):
        return None

    def get_dataset_args(self):
        args = {"norm2meters": self.norm2meters, "data_columns": list(self.columns), "delim": "tab",
                "wall_available": False, "scale": self.scale, "img_scaling": 0.05, "format": self.fmt}
        if self.framerate is not None:
            args["framerate"] = self.framerate
        h = self.homography()
        if h is not None:
            args["homography"] = h
        return args


class BiWi(Experiment):
    pass


class eth(BiWi):
    pass


class hotel(BiWi):
    pass


class univ(BiWi):
    pass


class zara1(BiWi):
    pass


class zara2(BiWi):
    pass


class stanford(Experiment):
    """Stanford Drone Dataset: pixel coordinates, 30 fps annotations, per-scene metres-per-pixel in H_SDD.txt."""
    columns, fmt, norm2meters, scale, framerate = SDD_COLUMNS, "pixel", True, True, 30

    def homography(self):
        return pd.read_csv(os.path.join(self.data_path, "H_SDD.txt"), delimiter="\t")


class gofp(Experiment):
    columns, fmt, norm2meters, scale, framerate = GOFP_COLUMNS, "pixel", True, True, 10

    def homography(self):
        return dict(GOFP_RATIOS)


def get(name):
    table = {c.__name__: c for c in (eth, hotel, univ, zara1, zara2, stanford, gofp)}
    key = name if name in table else name.lower()
    if key not in table:
        raise NotImplementedError("dataset '{}' (known: {})".format(name, ", ".join(sorted(table))))
    return table[key]()
