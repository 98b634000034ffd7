"""GPU: MultiGeneratorGAN.train() (reference loop /root/reference/mggan/abstract_train.py:114-168) on the launch mode the
benchmark measures: with --rng device the loop replays one captured HIP graph per batch shape (first appearance of a
shape: eager on static buffers; second: capture; then replays) and must train exactly like eager launches."""
import io
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(graph, peds, shapes=8, epochs=3, scenes=16, bs=4, extra=("--graph_pad", "off")):
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    cfg = get_parser().parse_args(["--num_gens", "2", "--rng", "device", "--graph", graph, "--graph_shapes", str(shapes),
                                   "--epochs", str(epochs), "--batch_size", str(bs), "--synthetic_scenes", str(scenes),
                                   "--synthetic_peds", str(peds), "--cache_device", "1", "--val_every", "1000",
                                   "--save_every", "1000"] + list(extra))
    torch.manual_seed(145325)
    np.random.seed(435346)
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    torch.cuda.manual_seed(77)
    g = torch.Generator().manual_seed(3)  # the loader shuffles with torch's global CPU generator
    torch.manual_seed(int(torch.randint(0, 2 ** 31, (1,), generator=g)))
    metrics = tr.train()
    torch.cuda.synchronize()
    return tr, metrics


@pytest.mark.parametrize("peds,shapes", [(3, 8), (0, 8), (0, 2)])
def test_train_replays_graphs_and_matches_eager(peds, shapes):
    """peds=3: one batch shape (one graph); peds=0: ragged scenes, a shape per batch of the epoch -- with a cache of two
    shapes most batches stay eager, and eager iterations run between replays of pinned graphs."""
    tr_g, m_g = _train("auto", peds, shapes)
    tr_e, m_e = _train("off", peds)
    ig = tr_g.iteration_graphs
    assert ig is not None and tr_e.iteration_graphs is None
    from mggan.data_utils import synthetic

    per_shape = {}  # batches of an epoch per distinct shape (ragged: the sizes of batch i are drawn with seed i)
    for i in range(4):
        k = tuple(synthetic.scene_sizes(4, peds or None, seed=i))
        per_shape[k] = per_shape.get(k, 0) + 1
    assert len(ig.entries) == min(len(per_shape), shapes)
    # a shape's first batch runs eagerly on the static buffers, every later one is a replay
    cached = {k for k, _ in [(e[0], 0) for e in ig.entries]}
    assert ig.replays == sum(3 * per_shape[k] - 1 for k in cached), (ig.replays, ig.eager)
    assert ig.replays + ig.eager + (12 - sum(3 * per_shape[k] for k in cached)) == 12
    assert sum(tr_g.epoch_iterations) == 12
    for k, v in m_e.items():
        assert np.isfinite(v) and np.isfinite(m_g[k]), k
        np.testing.assert_allclose(m_g[k], v, rtol=1e-5, atol=1e-7, err_msg=k)  # epoch mean of the logged losses
    for a, b in ((tr_g.G, tr_e.G), (tr_g.D, tr_e.D)):
        assert torch.equal(a._flat, b._flat)  # replays are bit-identical to eager launches


@pytest.mark.parametrize("bucket,shapes,graphs", [("quarter", 2, 2), ("pow2", 2, 1)])
def test_ragged_batches_replay_one_graph_per_bucket(bucket, shapes, graphs):
    """The reference loader's batches (trajectories_scene.py:40-78) carry a new tuple of scene sizes almost every time: 8
    batches per epoch of 8 scenes with 1-6 pedestrians, 22-29 pedestrians per batch.  Padded to shape buckets, a graph cache
    of TWO shapes replays more than 90 % of the iterations, and the run is bit-identical to eager launches on the same
    padded batches (--graph off --graph_pad on)."""
    extra = ["--graph_pad", "on", "--graph_bucket", bucket]
    tr_g, m_g = _train("auto", 0, shapes, epochs=4, scenes=64, bs=8, extra=extra)
    tr_e, m_e = _train("off", 0, shapes, epochs=4, scenes=64, bs=8, extra=extra)
    ig, ie = tr_g.iteration_graphs, tr_e.iteration_graphs
    total = sum(tr_g.epoch_iterations)
    assert total == 32 and len(ig.entries) == graphs and ig.padded == total and ie.padded == total
    assert ig.replays == total - graphs and ig.replays >= 0.9 * total and ig.eager == graphs
    assert ie.replays == 0 and ie.eager == total
    assert len(ig.history) == 4 and ig.history[-1] == (ig.replays, ig.eager, graphs)  # the replay rate, logged per epoch
    for k, v in m_e.items():
        assert np.isfinite(v) and np.isfinite(m_g[k]), k
        np.testing.assert_allclose(m_g[k], v, rtol=1e-5, atol=1e-7, err_msg=k)
    for a, b in ((tr_g.G, tr_e.G), (tr_g.D, tr_e.D)):
        assert torch.equal(a._flat, b._flat)  # replays of a bucket's graph == eager launches on the padded batches


def test_graph_follows_the_learning_rate_schedule():
    """The cosine schedule changes the learning rate every epoch; a captured iteration reads it from device memory."""
    tr, _ = _train("auto", 3, epochs=4)
    assert tr.iteration_graphs.replays == 4 * 4 - 1
    assert tr.optimizerG.lr < tr.optimizerG.base_lr and float(tr.optimizerG._lr_dev.cpu()) == tr.optimizerG.lr


def test_mgan_pm_target_and_gated_discriminator_steps_replay_like_eager():
    """Two configurations that used to launch eagerly: `--weighting_target mgan` (its 0.9 ** epoch is a device word now: a
    captured iteration follows the epochs) and `--num_gen_steps 2` (the discriminator step runs in every second iteration: the
    graph cache keeps one graph per shape with and one without it).  `--graph on` replays both, bit-identical to `--graph off`
    over three epochs."""
    # (--keep_gen_steps: the gating of abstract_train.py:136-150 only acts while epoch < keep_gen_steps, 0 by default)
    for extra in (["--weighting_target", "mgan"], ["--num_gen_steps", "2", "--keep_gen_steps", "100"]):
        ex = ["--graph_pad", "off"] + extra
        tr_g, m_g = _train("on", 3, extra=ex)
        tr_e, m_e = _train("off", 3, extra=ex)
        ig = tr_g.iteration_graphs
        n_graphs = 2 if "--num_gen_steps" in extra else 1
        assert ig is not None and len(ig.entries) == n_graphs and ig.eager == n_graphs and ig.replays == 12 - n_graphs, (
            extra, len(ig.entries), ig.eager, ig.replays)
        for k, v in m_e.items():
            np.testing.assert_allclose(m_g[k], v, rtol=1e-5, atol=1e-7, err_msg=k)
        for a, b in ((tr_g.G, tr_e.G), (tr_g.D, tr_e.D)):
            assert torch.equal(a._flat, b._flat), extra
        if "--weighting_target" in extra:
            assert abs(float(tr_g._pm_reg_dev.cpu()) - 0.9 ** 3) < 1e-7


def test_graph_on_needs_the_device_rng():
    from mggan.logging import Experiment
    from mggan.model.config import get_parser
    from mggan.model.model_factory import construct_model
    from mggan.model.train import PiNetMultiGeneratorGAN

    cfg = get_parser().parse_args(["--graph", "on", "--rng", "host", "--epochs", "1"])
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = construct_model(cfg)
    tr = PiNetMultiGeneratorGAN(G, D, cfg, Experiment(debug=True))
    with pytest.raises(ValueError):
        tr.train()


def test_bench_stdout_is_exactly_one_json_line():
    """The driver parses the ONE line bench.py prints: nothing else may reach stdout -- not the model summary of
    construct_model, not train()'s per-epoch lines (the floors and the train() legs run inside this command too)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--config", "c1",
                          "--also", "", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:5]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and not line["roofline"]["kernel"].startswith("mggan_")
