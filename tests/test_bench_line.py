"""bench.py prints ONE strict-JSON line of less than 4 KB (the driver keeps the last 8 KB of stdout; round 3's line
outgrew that and could not be parsed).  A canned, deliberately bloated result goes through the emit function."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _roof(i):
    return {"kernel": "decoder_bwd_mfma_kernel" + "x" * i, "entry": "mggan_decoder_rollout_bwd_fused", "traffic": 1.98e9,
            "mfma_util": 0.43, "launches_per_step": 1.0, "avg_launch_ms": 0.8, "timed": "graph replay (device-clock marks)",
            "standalone_ms": 0.79, "tflops": 49.0, "algorithmic_gbs": 2100.0, "bound": "mfma", "achieved": 49.0,
            "peak": 157.3, "unit": "TFLOP/s", "frac": 0.31, "note": "n" * 900}


def _canned():
    conf = {"config": "c2", "workload": "w" * 150, "b_per_gpu": 1280, "ms_per_step": 1.47, "value": 870000.0,
            "unit": "trajectories/s", "launch": "hipGraph replay of the whole iteration", "collective": None,
            "roofline": _roof(0), "roofline_top_kernels": [_roof(i) for i in range(8)],
            "iteration_frac_of_f32_peak": float("nan"), "launches_per_step": 98.0,
            "breakdown": [{"entry": "e" * 30, "kernel": "k" * 40, "ms_per_step": 0.1} for _ in range(14)]}
    return {"metric": "train-step trajectories/sec", "value": 870000.0, "unit": "trajectories/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 1.47, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "w" * 150, "b_per_gpu": 1280, "parallelism": "dp1", "rng": "device", "bn_sync": "global",
                       "launch": "hipGraph replay of the whole iteration", "collective": None,
                       "last_losses": {"l%d" % i: float("inf") for i in range(20)}},
            "roofline": _roof(0), "roofline_top_kernels": [_roof(i) for i in range(8)],
            "breakdown": conf["breakdown"], "configs": [conf, dict(conf, config="c3")],
            "c1_shaped": {"note": "z" * 500}, "train_loop": {"ms_per_step_by_epoch": [1.5] * 6},
            "cpu_baseline": {"value": 567.0123456, "unit": "trajectories/s", "cores": 16, "kind": "port", "mode": "block",
                             "sample": "s" * 600},
            "gpu_over_cpu_port": 1534.7,
            "collective_transports": [{"config": "c2", "peer-mapped": {"ms_per_step": 1.5, "value": 1.0, "launch": "x" * 80},
                                       "rccl-segments": {"ms_per_step": 1.9, "value": 0.8, "launch": "y" * 80}}]}


def test_compact_line_is_small_strict_json():
    B = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 8192  # the canned record is of the size that broke round 3
    text = B.compact_line(full)
    assert "\n" not in text and len(text) < 4096, len(text)
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # no NaN / Infinity tokens
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"] and "last_losses" not in line["config"]
    assert set(line["roofline"]) == set(B.ROOFLINE_KEYS) and "note" not in line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["cpu_baseline"]["kind"] == "port" and len(line["cpu_baseline"]["sample"]) <= 200
    assert [c["workload"] for c in line["configs"]] == ["c2", "c3"]
    assert line["configs"][0].get("iteration_frac_of_f32_peak") is None  # NaN never reaches the line
    assert "roofline_top_kernels" not in line and "breakdown" not in line and "c1_shaped" not in line
    assert not [k for k in line if k.startswith("gpu_over_cpu")]  # the ratio against the port stays in the detail file


def test_emit_writes_detail_file(tmp_path, monkeypatch):
    B = _bench()
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    text = B.emit(_canned())
    assert len(text) < 4096
    detail = json.loads((tmp_path / "bench_detail.json").read_text())
    assert len(detail["roofline_top_kernels"]) == 8 and "train_loop" in detail
    assert json.loads((tmp_path / "gpurun_out" / "bench_detail.json").read_text()) == detail
