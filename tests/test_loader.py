"""On-disk dataset loader against what the reference's loader produced from the same files
(tests/golden/golden_loader.npz, made by tests/golden/make_golden_loader.py): sequence extraction, scaling,
NaN padding of inactive pedestrians, image rescaling, the per-pedestrian crops, augmentation draws, collation."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def loader_golden(tmp_path_factory):
    g = dict(np.load(os.path.join(GOLDEN, "golden_loader.npz")))
    root = tmp_path_factory.mktemp("datasets")
    for k, v in g.items():
        if k.startswith("file/"):
            path = root / k[len("file/"):]
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_bytes(v.tobytes())
    os.environ["MGGAN_DATA_ROOT"] = str(root)
    yield g
    os.environ.pop("MGGAN_DATA_ROOT", None)


@pytest.mark.parametrize("name,small", [("eth", 0.5), ("gofp", 0.5), ("stanford", 0.7)])
@pytest.mark.parametrize("phase,aug", [("test", 0), ("train", 1)])
def test_dataset_matches_reference(loader_golden, name, small, phase, aug):
    from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval, seq_collate_scene

    g = loader_golden
    ds = TrajectoryDatasetEval(dataset_name=name, phase=phase, margin_in=16, margin_out=16, load_occupancy=False,
                               scaling_small=small, data_augmentation=aug)
    p = "{}/{}/".format(name, phase)
    np.testing.assert_array_equal(np.array(ds.seq_start_end), g[p + "seq_start_end"])
    np.testing.assert_array_equal(ds.ped_ids, g[p + "ped_ids"])
    assert list(ds.scene_list) == list(g[p + "scenes"])
    np.testing.assert_allclose(ds.trajectory, g[p + "trajectory"], rtol=1e-12, atol=0, equal_nan=True)
    np.random.seed(123)  # the augmentation draws come from numpy's global generator
    batch = seq_collate_scene([ds[i] for i in range(min(3, len(ds)))])
    assert batch["seq_start_end"] == g[p + "batch/seq_start_end"].tolist()
    for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
        np.testing.assert_allclose(batch[k].numpy(), g[p + "batch/" + k], rtol=1e-6, atol=1e-6, equal_nan=True, err_msg=k)
    # same decoded JPEG, same Pillow filters, same crop boxes: the crops agree to the bit
    np.testing.assert_array_equal(batch["features"].numpy(), g[p + "batch/features"])
    assert batch["features"].shape[1:] == (4, 33, 33)
    assert float(batch["features"][:, 3].sum()) == batch["features"].shape[0]  # one-hot centre channel


def test_get_dataloader_on_disk(loader_golden):
    from mggan.data_utils.data_loaders import get_dataloader

    loader = get_dataloader("eth", "test", batch_size=4)
    batch = next(iter(loader))
    b = batch["in_xy"].shape[1]
    assert batch["in_xy"].shape == (8, b, 2) and batch["gt_xy"].shape == (12, b, 2)
    assert batch["in_dxdy"].shape == (7, b, 2) and batch["features"].shape == (b, 4, 33, 33)
    assert batch["seq_start_end"][-1][1] == b


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["eth", "gofp", "stanford"])
def test_device_crops_match_host_crops(loader_golden, name):
    """mggan_crop_patches (scene images resident in HBM, one launch per batch) == the per-pedestrian PIL crops."""
    from mggan.data_utils.data_loaders import get_dataloader

    host = next(iter(get_dataloader(name, "test", batch_size=5)))
    dev = next(iter(get_dataloader(name, "test", batch_size=5, crop_device="cuda")))
    assert dev["features"].is_cuda and dev["features"].shape == host["features"].shape
    assert torch.equal(dev["features"].cpu(), host["features"])
    for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
        np.testing.assert_array_equal(dev[k].cpu().numpy(), host[k].numpy())  # (the whole batch arrives on the device)
    assert dev["seq_start_end"] == host["seq_start_end"]
    g = loader_golden["{}/test/batch/features".format(name)]
    n = min(len(g), dev["features"].shape[0])
    np.testing.assert_array_equal(dev["features"].cpu().numpy()[:n], g[:n])  # and == the reference's crops


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["eth", "gofp", "stanford"])
def test_device_crops_match_host_crops_with_training_augmentation(loader_golden, name):
    """phase="train", --augment 1 (the reference's default, trajectories_scene.py:276-357): the scene image is flipped,
    rotated (expand, nearest) and Lanczos-resized per item by Pillow on the host; mggan_crop_patches_aug computes every
    pedestrian's crop of that image directly from the resident un-augmented image -- the same bytes, and the reference's own."""
    from mggan.data_utils.data_loaders import get_dataloader

    np.random.seed(123)  # the augmentation draws (rotation angle, flip code) come from numpy's global generator
    host = next(iter(get_dataloader(name, "train", augment=True, batch_size=3)))
    np.random.seed(123)
    dev = next(iter(get_dataloader(name, "train", augment=True, batch_size=3, crop_device="cuda")))
    assert dev["features"].is_cuda and dev["features"].shape == host["features"].shape
    np.testing.assert_array_equal(dev["features"].cpu().numpy(), host["features"].numpy())
    for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
        np.testing.assert_array_equal(dev[k].cpu().numpy(), host[k].numpy())
    assert dev["seq_start_end"] == host["seq_start_end"]
    g = loader_golden["{}/train/batch/features".format(name)]
    assert len(g) == dev["features"].shape[0]
    np.testing.assert_array_equal(dev["features"].cpu().numpy(), g)  # == what the reference's loader produced


@pytest.mark.parametrize("name,small", [("eth", 0.5), ("gofp", 0.5), ("stanford", 0.7)])
@pytest.mark.parametrize("phase,aug", [("test", 0), ("train", 1)])
def test_batched_host_half_equals_per_item_path(loader_golden, name, small, phase, aug):
    """DeviceCropDataset.__getitems__ (the whole batch in one numpy pass: what the DataLoader calls) against its per-item
    __getitem__ + collate: the same trajectory tensors, crop centres and augmentation records to the bit, with the
    augmentation draws consumed in the same order -- and the trajectories are the reference loader's (CPU only: the device
    half is not touched)."""
    from mggan.data_utils.device_crops import DeviceCropDataset
    from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval

    ds = TrajectoryDatasetEval(dataset_name=name, phase=phase, margin_in=16, margin_out=16, load_occupancy=False,
                               scaling_small=small, data_augmentation=aug)
    dds = DeviceCropDataset(ds, "cpu")
    idx = list(range(min(3, len(ds))))
    np.random.seed(123)
    a = dds.collate_host([dds[i] for i in idx])
    state_a = np.random.get_state()[2]
    np.random.seed(123)
    b = dds.collate_host(dds.__getitems__(idx))
    assert np.random.get_state()[2] == state_a  # as many draws from numpy's global generator
    g = loader_golden
    p = "{}/{}/".format(name, phase)
    for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy"):
        np.testing.assert_array_equal(a[k].numpy(), b[k].numpy(), err_msg=k)  # (NaN == NaN here: pedestrians without ground truth)
        np.testing.assert_allclose(b[k].numpy(), g[p + "batch/" + k], rtol=1e-6, atol=1e-6, equal_nan=True, err_msg=k)
    assert a["seq_start_end"] == b["seq_start_end"] == g[p + "batch/seq_start_end"].tolist()
    assert a["scene_img"] == b["scene_img"] and (a["loss_mask"] is None) == (b["loss_mask"] is None)
    assert a["_crop_meta"][0] == b["_crop_meta"][0]
    for x, y in zip(a["_crop_meta"][1:], b["_crop_meta"][1:]):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("phase,aug", [("test", 0), ("train", 1)])
def test_slot_round_trip_of_a_worker_batch(loader_golden, phase, aug):
    """--workers N: a worker writes the host half of a batch (and its launch plan) into a shared-memory slot and sends a
    skeleton; the parent's rebuilt batch equals the in-process one, array for array (CPU only)."""
    import pickle

    from mggan.data_utils.device_crops import DeviceCropDataset
    from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval

    ds = TrajectoryDatasetEval(dataset_name="eth", phase=phase, margin_in=16, margin_out=16, load_occupancy=False,
                               scaling_small=0.5, data_augmentation=aug)
    dds = DeviceCropDataset(ds, "cpu")
    idx = list(range(min(4, len(ds))))
    np.random.seed(5)
    ref = dds.__getitems__(idx)[0].batch
    plan = dds.plan(ref["_crop_meta"])
    np.random.seed(5)
    slot = np.zeros(1 << 20, np.uint8)
    skel = pickle.loads(pickle.dumps(dds.to_slot(dds.__getitems__(idx)[0].batch, slot)))  # (what crosses the pipe)
    assert len(pickle.dumps(skel)) < 64 << 10
    got = DeviceCropDataset.from_slot(skel, slot)
    for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy", "size", "_traj_base"):
        np.testing.assert_array_equal(got[k].numpy(), ref[k].numpy(), err_msg=k)
    assert got["seq_start_end"] == ref["seq_start_end"] and got["scene_img"] == ref["scene_img"]
    assert (got["loss_mask"] is None) == (ref["loss_mask"] is None) and len(got["occupancy"]) == len(ref["occupancy"])
    gp = got["_crop_plan"]
    assert {k: v for k, v in gp.items() if k != "arrays"} == {k: v for k, v in plan.items() if k != "arrays"}
    for x, y in zip(gp["arrays"], plan["arrays"]):
        assert x.dtype == y.dtype
        np.testing.assert_array_equal(x, y)
    assert dds.to_slot(dds.__getitems__(idx)[0].batch, np.zeros(64, np.uint8)) is None  # too small a slot: the parent's job


@pytest.mark.gpu
@pytest.mark.parametrize("phase,aug", [("test", False), ("train", True)])
def test_loader_workers_hand_over_the_same_batches(loader_golden, phase, aug, monkeypatch):
    """get_dataloader(..., workers=2, crop_device=...): two forked processes do the host halves, the batches arrive in
    sampler order and equal the in-process loader's (the augmentation draws are pinned to one rotation + flip so that the
    processes' own generators do not matter), over two epochs (the workers persist)."""
    from mggan.data_utils import trajectories_scene as TS
    from mggan.data_utils.data_loaders import get_dataloader

    if aug:
        monkeypatch.setattr(TS.TrajectoryDatasetEval, "augmentation", lambda self: (0.7, 1))
    one = get_dataloader("eth", phase, augment=aug, batch_size=2, crop_device="cuda")
    two = get_dataloader("eth", phase, augment=aug, batch_size=2, crop_device="cuda", workers=2)
    try:
        for _ in range(2):
            n = 0
            for a, b in zip(one, two):
                n += 1
                for k in ("in_xy", "gt_xy", "in_dxdy", "gt_dxdy", "features"):
                    assert b[k].is_cuda
                    np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), err_msg=k)
                assert a["seq_start_end"] == b["seq_start_end"] and a["scene_img"] == b["scene_img"]
            assert n == len(one) == len(two) and n >= 2
    finally:
        two.close()


def test_slot_worker_without_the_tables_leaves_the_plan_to_the_parent(loader_golden):
    """A loader worker whose copy of the Lanczos table pool lacks a size (datasets with too many scene sizes to build every
    table ahead of time) sends the crop records unresolved; the parent resolves them against ITS pool: the same plan."""
    from mggan.data_utils.device_crops import DeviceCropDataset
    from mggan.data_utils.trajectories_scene import TrajectoryDatasetEval

    ds = TrajectoryDatasetEval(dataset_name="eth", phase="train", margin_in=16, margin_out=16, load_occupancy=False,
                               scaling_small=0.5, data_augmentation=1)
    parent = DeviceCropDataset(ds, "cpu")
    worker = DeviceCropDataset(ds, "cpu")
    worker._pool, worker._pool_idx, worker._pool_len, worker._frozen = [], {}, 0, True  # (forked before any table existed)
    idx = list(range(min(3, len(ds))))
    np.random.seed(9)
    want = parent.plan(parent.__getitems__(idx)[0].batch["_crop_meta"])
    np.random.seed(9)
    slot = np.zeros(1 << 20, np.uint8)
    skel = worker.to_slot(worker.__getitems__(idx)[0].batch, slot)
    assert "_crop_plan" not in skel and "_crop_meta" in skel
    got = DeviceCropDataset.from_slot(skel, slot)
    plan = parent.plan(got["_crop_meta"])
    assert {k: v for k, v in plan.items() if k != "arrays"} == {k: v for k, v in want.items() if k != "arrays"}
    for x, y in zip(plan["arrays"], want["arrays"]):
        np.testing.assert_array_equal(x, y)
