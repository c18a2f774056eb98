"""CPU tests of the h5py-free Keras-3 checkpoint interchange (adunet_amd/hdf5_min.py, keras_archive.py).

Round trips through the module and checks of the layout rules restated from Keras 3's saving_lib (store names by class counter
in `model.layers` order, variable order per class): Keras is absent, so the STORE NAMING is unpinned.  The HDF5 container
underneath is pinned against real libhdf5 in tests/test_against_second_interpreter.py.  The GPU side (a model's weights through an archive and back) is tests/test_checkpoint_gpu.py."""
import json
import zipfile
from collections import OrderedDict

import numpy as np
import pytest
import torch

from adunet_amd import hdf5_min as H
from adunet_amd import keras_archive as K
from adunet_amd.model import build_super_resolution_unet
from adunet_amd.seg_model import build_adaptive_depth_unet, build_unet


class HostWeights:
    """A model's graph (built without a GPU) with host-side weights: what keras_archive needs of a model."""

    def __init__(self, model, seed=0):
        self.m = model
        self.layers, self.index, self.name, self.convs = model.layers, model.index, model.name, model.convs
        self.state_index = getattr(model, "state_index", {})
        self.scale = getattr(model, "scale", 0.0)
        rng = np.random.default_rng(seed)
        self.w = OrderedDict((n, rng.standard_normal(s).astype(np.float32)) for n, (_, s) in list(self.index.items()) + list(self.state_index.items()))
        self.loaded = None

    def get_weights(self):
        return self.w

    def set_weights(self, w):
        self.loaded = w


def same_tree(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and set(a) == set(b) and all(same_tree(a[k], b[k]) for k in a)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def test_hdf5_subset_round_trip():
    rng = np.random.default_rng(0)
    tree = {"layers": {f"conv2d_{i}": {"vars": {"0": rng.standard_normal((3, 3, 4, 8)).astype(np.float32),
                                                "1": rng.standard_normal(8).astype(np.float32)}} for i in range(40)},
            "vars": {}, "optimizer": {"vars": {"0": np.array(7, dtype=np.int64), "1": rng.standard_normal((2, 5)),
                                               "2": np.arange(6, dtype=np.int32).reshape(2, 3)}}}
    tree["layers"]["activation"] = {"vars": {}}                      # an empty group, as Keras leaves for layers without variables
    data = H.write_file(tree)
    assert data[:8] == H.SIGNATURE and len(data) % 8 == 0
    assert same_tree(tree, H.read_file(data))                        # 41 links in one group: six symbol-table nodes under one tree node
    x = rng.standard_normal((4, 6)).astype(">f4")[:, ::2]            # big-endian, non-contiguous: stored by value, little endian
    y = H.read_file(H.write_file({"a": x}))["a"]
    assert y.dtype == np.dtype("<f4") and np.array_equal(y, x.astype("<f4"))
    with pytest.raises(H.Hdf5Unsupported, match="not an HDF5 file"):
        H.read_file(b"PK\x03\x04 definitely a zip")
    with pytest.raises(H.Hdf5Unsupported, match="dtype"):
        H.write_file({"s": np.array(["text"])})
    with pytest.raises(H.Hdf5Unsupported, match="two-level"):
        H.write_file({str(i): np.zeros(1, np.float32) for i in range(257)})
    newer = bytearray(data)
    newer[8] = 2                                                     # a libver='latest' superblock must be refused, not misread
    with pytest.raises(H.Hdf5Unsupported, match="superblock version 2"):
        H.read_file(bytes(newer))


def test_store_names_follow_keras_container_numbering():
    assert [K.to_snake_case(n) for n in ("Conv2D", "LayerNormalization", "InputLayer", "ResizeByScale", "ResizeToMatch",
                                         "ClippedResidualAdd", "MaxPooling2D", "UpSampling2D", "Conv2DTranspose",
                                         "BatchNormalization", "Concatenate", "Activation")] == \
        ["conv2d", "layer_normalization", "input_layer", "resize_by_scale", "resize_to_match", "clipped_residual_add",
         "max_pooling2d", "up_sampling2d", "conv2d_transpose", "batch_normalization", "concatenate", "activation"]
    model, _ = build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=torch.float32)
    layout = K.store_layout(model)
    stores = [s for s, _ in layout]
    assert len(stores) == len(set(stores)) == len(model.layers)
    by_store = dict(layout)
    assert by_store["conv2d"] == ["conv2d/kernel", "conv2d/bias"]
    # the store ignores layer NAMES: residual_rgb is the last Conv2D of model.layers, numbered like the others
    n_conv = sum(r.type == "Conv2D" for r in model.layers)
    assert by_store[f"conv2d_{n_conv - 1}"] == ["residual_rgb/kernel", "residual_rgb/bias"]
    assert by_store["layer_normalization_1"] == ["layer_normalization_1/gamma", "layer_normalization_1/beta"]
    assert by_store["resize_by_scale"] == [] and by_store["resize_to_match"] == [] and by_store["clipped_residual_add"] == []
    assert "resize_by_scale_1" not in by_store                       # the shared enc_down layer is ONE layer of model.layers
    seg = build_adaptive_depth_unet(32, 32, 2, dtype=torch.float32)
    bn = dict(K.store_layout(seg))["batch_normalization"]
    assert [v.split("/")[1] for v in bn] == ["gamma", "beta", "moving_mean", "moving_variance"]
    ct = dict(K.store_layout(build_unet(32, 1, 32, 2, dtype=torch.float32)))["conv2d_transpose"]
    assert [v.split("/")[1] for v in ct] == ["kernel", "bias"]


@pytest.mark.parametrize("kind", ["sr", "bn", "ln_convT"])
def test_archive_round_trip_and_contents(tmp_path, kind):
    model = {"sr": lambda: build_super_resolution_unet(0.6, depth_override=2, input_size=40, dtype=torch.float32)[0],
             "bn": lambda: build_adaptive_depth_unet(32, 32, 2, dtype=torch.float32),
             "ln_convT": lambda: build_unet(32, 1, 32, 2, dtype=torch.float32)}[kind]()
    src = HostWeights(model, seed=3)
    path = tmp_path / "m.keras"
    K.save_keras(src, path)
    with zipfile.ZipFile(path) as z:
        assert sorted(z.namelist()) == ["config.json", "metadata.json", "model.weights.h5"]
        cfg = json.loads(z.read("config.json"))
        tree = H.read_file(z.read("model.weights.h5"))
    assert set(tree) == {"layers", "vars"} and tree["vars"] == {} and set(tree["layers"]) == {s for s, _ in K.store_layout(model)}
    assert all(set(g) == {"vars"} for g in tree["layers"].values())
    info = K.describe(path)
    assert info["metadata"]["keras_version"] == "3.3.3" and info["model_name"] == model.name and info["layers"] == len(model.layers)
    assert cfg["config"]["layers"][0]["class_name"] == "InputLayer" and cfg["config"]["output_layers"] == [[model.layers[-1].name, 0, 0]]
    if kind == "sr":
        assert info["custom"] == ["resize>ResizeByScale", "resize>ResizeToMatch", "utils>ClippedResidualAdd"]
        up = next(l for l in cfg["config"]["layers"] if l["class_name"] == "ResizeToMatch")
        assert len(up["inbound_nodes"][0]["args"][0]) == 2           # [x, skip] arrive as one list argument
    dst = HostWeights(model, seed=99)
    K.load_into(dst, path)
    assert set(dst.loaded) == set(src.w) and all(np.array_equal(dst.loaded[k], src.w[k]) for k in src.w)
    # a bare weights file is the same container without the zip
    K.save_weights_h5(src, tmp_path / "m.weights.h5")
    assert same_tree(K.read_weights(tmp_path / "m.weights.h5"), tree)


def test_archives_of_another_architecture_are_refused(tmp_path):
    a = HostWeights(build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=torch.float32)[0])
    K.save_keras(a, tmp_path / "d2.keras")
    deeper = HostWeights(build_super_resolution_unet(0.5, depth_override=3, input_size=32, dtype=torch.float32)[0])
    with pytest.raises(ValueError, match="no /layers/|shape"):
        K.load_into(deeper, tmp_path / "d2.keras")
    wider = HostWeights(build_super_resolution_unet(0.5, base_channels=32, depth_override=2, input_size=32, dtype=torch.float32)[0])
    with pytest.raises(ValueError, match="shape"):
        K.load_into(wider, tmp_path / "d2.keras")
    with zipfile.ZipFile(tmp_path / "empty.keras", "w") as z:
        z.writestr("config.json", "{}")
    with pytest.raises(ValueError, match="without model.weights.h5"):
        K.read_weights(tmp_path / "empty.keras")
    (tmp_path / "legacy.h5").write_bytes(H.write_file({"model_weights": {}}))
    with pytest.raises(ValueError, match="no /layers group"):
        K.load_into(a, tmp_path / "legacy.h5")


def test_the_reader_survives_corrupt_files():
    """A `.keras` / `.h5` file comes from outside.  Truncations, flipped bytes, addresses redirected to the file's start, end or
    beyond it (cycles, out-of-file pointers) and zeroed spans must each either read or raise Hdf5Unsupported -- no other
    exception type, no endless walk of a cyclic B-tree (every address is bounds-checked, trees and links are checked for cycles)."""
    import random
    import time
    rng = np.random.default_rng(0)
    good = H.write_file({"layers": {f"c{i}": {"vars": {"0": rng.standard_normal((3, 4)).astype(np.float32)}} for i in range(20)},
                         "vars": {}})
    random.seed(1)
    refused = 0
    t0 = time.time()
    for trial in range(1500):
        b = bytearray(good)
        mode = trial % 4
        if mode == 0:
            b = b[:random.randrange(0, len(b))]
        elif mode == 1:
            for _ in range(random.randint(1, 8)):
                b[random.randrange(len(b))] = random.randrange(256)
        elif mode == 2:
            o = random.randrange(8, len(b) - 8) & ~7
            b[o:o + 8] = random.choice([0, 96, len(b) - 8, len(b) + 100, random.randrange(len(b)), H.UNDEF]).to_bytes(8, "little")
        else:
            o = random.randrange(len(b))
            b[o:o + random.randint(1, 64)] = b"\0" * min(64, len(b) - o)
        try:
            H.read_file(bytes(b))
        except H.Hdf5Unsupported:
            refused += 1
    assert refused > 300 and time.time() - t0 < 30
    for junk in (b"", b"\x89HDF\r\n\x1a\n", b"\x89HDF\r\n\x1a\n" + b"\xff" * 200):
        with pytest.raises(H.Hdf5Unsupported):
            H.read_file(junk)
