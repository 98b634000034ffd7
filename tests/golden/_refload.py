"""Import the read-only reference (/root/reference) in THIS container only.

Used solely by make_golden.py to generate golden vectors; never on the GPU box.
Three third-party packages the reference imports are absent here (test_tube,
cv2, shapely) -> placeholder modules are injected into sys.modules.
"""
import sys
import types
import argparse
import tempfile

REF = "/root/reference"


def install_stubs():
    sys.dont_write_bytecode = True
    if "test_tube" not in sys.modules:
        tt = types.ModuleType("test_tube")

        class HyperOptArgumentParser(argparse.ArgumentParser):
            def __init__(self, *a, strategy=None, **kw):
                super().__init__(*a, **kw)

            def opt_list(self, *a, options=None, tunable=None, **kw):
                return self.add_argument(*a, **kw)

        class Experiment:
            def __init__(self, *a, name="x", version=0, **kw):
                self.name, self.version = name, version
                self._d = tempfile.mkdtemp(prefix="tt_")

            def get_data_path(self, name, version):
                return self._d

            def log(self, *a, **kw):
                pass

            def save(self):
                pass

            def argparse(self, *a):
                pass

        tt.HyperOptArgumentParser = HyperOptArgumentParser
        tt.Experiment = Experiment
        sys.modules["test_tube"] = tt
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "shapely" not in sys.modules:
        sh = types.ModuleType("shapely")
        geo = types.ModuleType("shapely.geometry")
        ops = types.ModuleType("shapely.ops")
        geo.Polygon = geo.MultiPolygon = object
        ops.unary_union = lambda *a, **k: None
        sh.geometry, sh.ops = geo, ops
        sys.modules.update({"shapely": sh, "shapely.geometry": geo, "shapely.ops": ops})
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_reference():
    install_stubs()
    import mggan.model.train as ref_train  # noqa
    import mggan.model.config as ref_config  # noqa
    return ref_train, ref_config
