"""The training augmentation's image path (/root/reference/mggan/data_utils/trajectories_scene.py:276-357): the numpy
oracle (oracle/pil_crops_oracle.py) and the integer tables of mggan/data_utils/aug_geometry.py against the installed Pillow --
whole images, bit for bit -- and, on the GPU, the device crop kernel (csrc/crop.hip: crop_patches_aug_kernel) against both."""
import numpy as np
import pytest
from PIL import Image

import pil_crops_oracle as PO
from mggan.data_utils import aug_geometry as AG


def _image(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 7 + 2, w // 7 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR)).copy()  # structured, not white noise
    img[::13, ::11] = rng.integers(0, 256, img[::13, ::11].shape, dtype=np.uint8)
    return img


@pytest.mark.parametrize("w,h,alpha", [(97, 64, 0.3), (64, 97, 2.0), (120, 120, 4.4), (33, 200, 6.1), (211, 157, 1.5707), (80, 60, 3.3)])
def test_rotate_expand_is_pillow_nearest_affine(w, h, alpha):
    img = _image(w, h, 1)
    want = np.asarray(Image.fromarray(img).rotate(alpha / np.pi * 180, expand=True))
    got = PO.rotate_expand(img, alpha)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("w,h,f", [(200, 150, 0.1), (333, 257, 0.1), (640, 480, 1 / 14.0), (150, 90, 0.5), (90, 200, 0.31)])
def test_resize_lanczos_is_pillow_8bit_two_pass(w, h, f):
    img = _image(w, h, 2)
    size = AG.small_size(w, h, f)
    want = np.asarray(Image.fromarray(img).resize(size, Image.LANCZOS))
    np.testing.assert_array_equal(PO.resize_lanczos(img, size), want)


@pytest.mark.parametrize("code", [0, 1, 2])
def test_whole_augmented_small_image(code):
    img = _image(260, 190, 3)
    rng = np.random.default_rng(5 + code)
    for _ in range(3):
        alpha = float(rng.random() * 2 * np.pi)
        pil = Image.fromarray(img)
        if code == 1:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        elif code == 2:
            pil = pil.transpose(Image.FLIP_TOP_BOTTOM)
        pil = pil.rotate(alpha / np.pi * 180, expand=True)
        want = np.asarray(pil.resize(AG.small_size(pil.width, pil.height, 0.1), Image.LANCZOS))
        np.testing.assert_array_equal(PO.augmented_small_image(img, code, alpha, 0.1), want)
        box = (-5, want.shape[0] - 20, 28, want.shape[0] + 13)
        np.testing.assert_array_equal(PO.crop(want, box), np.asarray(Image.fromarray(want).crop(box)))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,f", [(640, 480, 0.1), (333, 517, 1 / 14.0), (200, 150, 0.31), (1100, 900, 1 / 14.0)])
def test_device_kernel_matches_pillow_crops(w, h, f):
    """mggan_crop_patches_aug on its own: random flips / angles / crop centres (windows that leave the image on every side
    included) against crops Pillow cuts out of the image it transformed itself -- and the oracle agrees with both."""
    import torch

    from mggan.data_utils import device_crops as DC
    from mggan.hip.lib import lib

    img = _image(w, h, 7)
    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)

    class _DS:  # the few attributes DeviceCropDataset reads
        data_augmentation, phase, images = 1, "train", {"s": {"scaled_image": Image.fromarray(img)}}
        img_scaling, scaling_small, margin_in = f, 1.0, 16

    dds = DC.DeviceCropDataset(_DS, dev)
    items, ped_item, ctr, want = [], [], [], []
    for i in range(6):
        alpha, code = (0.0, 0) if i == 0 else (float(rng.random() * 2 * np.pi), int(rng.integers(0, 3)))
        items.append(dds._aug_item("s", alpha, code))
        pil = Image.fromarray(img)
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT) if code == 1 else pil.transpose(Image.FLIP_TOP_BOTTOM) if code == 2 else pil
        pil = pil.rotate(alpha / np.pi * 180, expand=True)
        small = pil.resize(AG.small_size(pil.width, pil.height, f), Image.LANCZOS)
        np.testing.assert_array_equal(PO.augmented_small_image(img, code, alpha, f), np.asarray(small))
        for j in range(7):
            xc = int(rng.integers(-20, small.width + 20)) if j else small.width // 2
            yc = int(rng.integers(-20, small.height + 20)) if j else small.height // 2
            ped_item.append(i)
            ctr.append((xc, yc))
            want.append(np.asarray(small.crop((xc - 16, yc - 16, xc + 17, yc + 17))))
    pool = torch.from_numpy(np.concatenate(dds._pool)).to(dev)
    it = torch.from_numpy(np.stack(items)).to(dev)
    pi = torch.tensor(ped_item, dtype=torch.int32, device=dev)
    ct = torch.tensor(ctr, dtype=torch.int32, device=dev)
    out = torch.full((len(ped_item), 4, 33, 33), float("nan"), device=dev)
    lib.mggan_crop_patches_aug(dds.atlas.data_ptr(), it.data_ptr(), pool.data_ptr(), pi.data_ptr(), ct.data_ptr(), len(ped_item), 16,
                               int(np.stack(items)[:, 16].max()), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = np.stack(want).astype(np.float64)
    np.testing.assert_array_equal(got[:, :3], (-1 + ref * 2.0 / 256).astype(np.float32).transpose(0, 3, 1, 2))
    centre = np.zeros((33, 33), np.float32)
    centre[16, 16] = 1
    assert all(np.array_equal(got[p, 3], centre) for p in range(len(ped_item)))


@pytest.mark.gpu
def test_whole_image_tiles_equal_per_crop_windows(monkeypatch):
    """With many pedestrians per item the loader computes each item's whole resized image once (33 x 33 tiles,
    mggan_aug_small_images) and cuts plain windows from it; with few it computes the windows directly.  Same bits both ways, and
    Pillow's."""
    import torch

    from mggan.data_utils import device_crops as DC

    img = _image(420, 300, 9)
    rng = np.random.default_rng(3)
    dev = torch.device("cuda", 0)

    class _DS:
        data_augmentation, phase, images = 1, "train", {"s": {"scaled_image": Image.fromarray(img)}}
        img_scaling, scaling_small, margin_in = 0.1, 1.0, 16

    dds = DC.DeviceCropDataset(_DS, dev)
    recs, ped_item, ctr, want = [], [], [], []
    for i in range(2):
        alpha, code = float(rng.random() * 2 * np.pi), int(rng.integers(0, 3))
        recs.append(dds._aug_item("s", alpha, code, resolve=False))
        pil = Image.fromarray(img)
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT) if code == 1 else pil.transpose(Image.FLIP_TOP_BOTTOM) if code == 2 else pil
        pil = pil.rotate(alpha / np.pi * 180, expand=True)
        small = pil.resize(AG.small_size(pil.width, pil.height, 0.1), Image.LANCZOS)
        for _ in range(30):
            xc, yc = int(rng.integers(-10, small.width + 10)), int(rng.integers(-10, small.height + 10))
            ped_item.append(i)
            ctr.append((xc, yc))
            want.append(np.asarray(small.crop((xc - 16, yc - 16, xc + 17, yc + 17))))
    meta = ("aug", np.stack(recs), np.array(ped_item, np.int32), np.array(ctr, np.int32))
    outs = {}
    for tiles in (True, False):
        monkeypatch.setattr(DC, "TILE_MODE", tiles)
        b = dds.finish({"_crop_meta": meta, "seq_start_end": [[0, 30], [30, 60]]})
        torch.cuda.synchronize()
        outs[tiles] = b["features"].cpu().numpy()
    assert np.array_equal(outs[True], outs[False])
    ref = np.stack(want).astype(np.float64)
    np.testing.assert_array_equal(outs[True][:, :3], (-1 + ref * 2.0 / 256).astype(np.float32).transpose(0, 3, 1, 2))
