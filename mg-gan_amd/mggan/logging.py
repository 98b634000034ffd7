"""Minimal experiment logger with the surface the reference uses from test_tube.Experiment
(abstract_train.py:27,35-37,194,201; train.py:678-690): name, version, get_data_path, log, save,
argparse.  Writes metrics.csv and meta_tags.csv under <save_dir>/<name>/version_<v>/."""
import csv
import os


class Experiment:
    def __init__(self, save_dir=".", name="default", debug=False, version=0, **kwargs):
        self.save_dir, self.name, self.debug, self.version = str(save_dir), name, debug, version
        self.metrics, self.tags = [], {}
        if not debug:
            os.makedirs(self.get_data_path(name, version), exist_ok=True)

    def get_data_path(self, name, version):
        return os.path.join(self.save_dir, name, "version_{}".format(version))

    def argparse(self, args):
        self.tags.update(vars(args))

    def log(self, metrics, epoch=None):
        row = dict(metrics)
        if epoch is not None:
            row["epoch"] = epoch
        self.metrics.append(row)

    def save(self):
        if self.debug:
            return
        d = self.get_data_path(self.name, self.version)
        with open(os.path.join(d, "meta_tags.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["key", "value"])
            for k, v in self.tags.items():
                w.writerow([k, v])
        keys = sorted({k for r in self.metrics for k in r})
        with open(os.path.join(d, "metrics.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            for r in self.metrics:
                w.writerow(r)
