"""Pre-processing: oracle vs the reference's golden vectors (CPU), HIP kernel vs oracle (gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO

MEAN, STD = [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0]
TAGS = ("small", "capped", "vg")


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "preprocess.npz"))


def raws_of(g, tag):
    return [np.random.Generator(np.random.PCG64(7700 + i)).uniform(0, 255, (int(h), int(w), 3)).astype(np.float32)
            for i, (h, w) in enumerate(g[f"{tag}/raw_shapes"])]


def check(images, sizes, scales, g, tag, tol):
    np.testing.assert_array_equal(np.asarray(sizes), g[f"{tag}/sizes"])
    np.testing.assert_allclose(np.asarray(scales), g[f"{tag}/scales_yx"], rtol=1e-7)
    images = np.asarray(images)
    if tag == "vg":
        assert list(images.shape) == g["vg/images_shape"].tolist()
        ref = g["vg/images_crop"]
        assert np.abs(images[:, :, 100:164, 200:264] - ref).max() / np.abs(ref).max() <= tol
        np.testing.assert_allclose(images.astype(np.float64).sum(axis=(2, 3)), g["vg/images_sum"], rtol=1e-5)
    else:
        ref = g[f"{tag}/images"]
        assert images.shape == ref.shape
        assert np.abs(images - ref).max() / np.abs(ref).max() <= tol


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_vs_reference_golden(golden, tag):
    mn, mx = golden[f"{tag}/minmax"].tolist()
    images, sizes, scales = PO.preprocess(raws_of(golden, tag), mn, mx, MEAN, STD)
    check(images.numpy(), sizes.numpy(), scales.numpy(), golden, tag, tol=1e-6)


def test_size_rule_known_answers():
    assert PO.resized_hw(375, 500, 800, 1333) == (800, 1067)
    assert PO.resized_hw(500, 375, 800, 1333) == (1067, 800)
    assert PO.resized_hw(300, 1000, 800, 1333) == (400, 1333)       # capped by the long edge
    assert PO.resized_hw(800, 800, 800, 1333) == (800, 800)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_preprocess_vs_golden(golden, tag):
    from vltk_amd.config import vg_c4_config_dict, Config
    from vltk_amd.preprocess import Preprocess
    mn, mx = golden[f"{tag}/minmax"].tolist()
    d = vg_c4_config_dict()
    d["input"]["min_size_test"], d["input"]["max_size_test"] = mn, mx
    pre = Preprocess(Config(d))
    raws = [torch.from_numpy(r).cuda() for r in raws_of(golden, tag)]
    ids, images, sizes, scales = pre(raws, list(range(len(raws))))
    assert ids == list(range(len(raws))) and images.is_cuda
    check(images.cpu().numpy(), sizes.cpu().numpy(), scales.cpu().numpy(), golden, tag, tol=2e-6)
