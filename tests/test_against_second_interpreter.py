"""Independent implementations from the image's SECOND interpreter as checkers (CPU): real libhdf5 for the h5py-free HDF5
subset (adunet_amd/hdf5_min.py), both ways, and scikit-image for the oracle's SSIM.

h5py is not importable in the interpreter the product and the tests run in, and it must not become a dependency.  The image
does carry a second interpreter, /opt/conda/bin/python3.9 (Anaconda), with h5py 3.3.0 on libhdf5 1.10.6.  It is used here ONLY
as an independent implementation in a subprocess with a clean environment:

  * writer pinned: a file from hdf5_min.write_file is opened by real h5py, which must find every group and read every dataset
    (values, dtype, shape incl. rank 0) identically;
  * reader pinned: real h5py (default libver, as Keras opens its weight files) writes the Keras-3 store layout -- more than 256
    links in one group (a two-level group B-tree), empty `vars` groups, attributes on groups and datasets (object-header
    continuation blocks), float16 / float64 / rank-0 int64 -- and hdf5_min.read_file must return exactly that tree;
  * files outside the subset (libver='latest', chunked + gzip, big-endian) must be REFUSED by name, never misread.

What this does NOT pin: Keras' own naming of the store (`layers/<snake-cased class + counter>/vars/<i>`, keras_archive.py) --
Keras is absent from both interpreters.  Skipped where the second interpreter or its h5py is missing."""
import os
import subprocess
import sys

import numpy as np
import pytest

from adunet_amd import hdf5_min as H

CONDA_PY = "/opt/conda/bin/python3.9"


def run_h5py(code: str, *args: str) -> str:
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")}
    res = subprocess.run([CONDA_PY, "-c", code, *args], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


@pytest.fixture(scope="module")
def h5py_version():
    if not os.path.exists(CONDA_PY):
        pytest.skip(f"{CONDA_PY} is not in this image")
    try:
        out = run_h5py("import h5py; print(h5py.__version__, h5py.version.hdf5_version)")
    except (AssertionError, OSError, subprocess.TimeoutExpired) as exc:
        pytest.skip(f"no usable h5py in {CONDA_PY}: {exc}")
    return out.split()


def flat(tree, prefix=""):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from flat(v, prefix + k + "/")
        else:
            yield prefix + k, v


def groups(tree, prefix=""):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield prefix + k
            yield from groups(v, prefix + k + "/")


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


READ_WITH_H5PY = """
import sys, h5py, numpy as np
out, grp = {}, []
def visit(name, obj):
    if isinstance(obj, h5py.Dataset): out[name] = obj[()]
    else: grp.append(name)
with h5py.File(sys.argv[1], "r") as f:
    f.visititems(visit)
np.savez(sys.argv[2], __groups__=np.array(grp), **{k.replace("/", "|"): v for k, v in out.items()})
"""


def test_real_h5py_reads_what_the_writer_writes(tmp_path, h5py_version):
    rng = np.random.default_rng(0)
    tree = {"layers": {("conv2d" if i == 0 else f"conv2d_{i}"): {"vars": {"0": rng.standard_normal((3, 3, 4, 8)).astype(np.float32),
                                                                               "1": rng.standard_normal(8).astype(np.float32)}}
                       for i in range(40)},
            "vars": {},
            "optimizer": {"vars": {"0": np.array(7, dtype=np.int64), "1": rng.standard_normal((2, 5)),
                                   "2": np.arange(6, dtype=np.int32).reshape(2, 3)}}}
    tree["layers"]["activation"] = {"vars": {}}
    path, back = tmp_path / "mine.h5", tmp_path / "back.npz"
    path.write_bytes(H.write_file(tree))
    run_h5py(READ_WITH_H5PY, str(path), str(back))
    z = np.load(back)
    got = {k.replace("|", "/"): z[k] for k in z.files if k != "__groups__"}
    want = dict(flat(tree))
    assert set(got) == set(want) and all(same(got[k], want[k]) for k in want)
    assert set(z["__groups__"].tolist()) == set(groups(tree))            # incl. the empty `vars` groups
    assert got["optimizer/vars/0"].shape == ()                             # a rank-0 dataset stays rank 0


WRITE_WITH_H5PY = """
import sys, h5py, numpy as np
rng = np.random.default_rng(1)
ref = {}
with h5py.File(sys.argv[1], "w") as f:                # default libver: what Keras' H5IOStore gets from h5py.File(path, "w")
    f.create_group("vars")
    layers = f.create_group("layers")
    for i in range(300):                              # > 256 links in one group: a two-level group B-tree
        name = "conv2d" if i == 0 else "conv2d_%d" % i
        g = layers.create_group(name).create_group("vars")
        k = rng.standard_normal((3, 3, 2, 4)).astype(np.float32); b = rng.standard_normal(4).astype(np.float32)
        g["0"] = k; g["1"] = b
        ref["layers/%s/vars/0" % name] = k; ref["layers/%s/vars/1" % name] = b
    layers.create_group("activation").create_group("vars")
    bn = layers.create_group("batch_normalization").create_group("vars")
    for j in range(4):
        v = rng.standard_normal(16).astype(np.float32); bn[str(j)] = v; ref["layers/batch_normalization/vars/%d" % j] = v
    opt = f.create_group("optimizer").create_group("vars")
    opt["0"] = np.int64(12345); ref["optimizer/vars/0"] = np.asarray(np.int64(12345))
    opt["1"] = rng.standard_normal((5, 7)); ref["optimizer/vars/1"] = opt["1"][()]
    opt["2"] = rng.standard_normal(9).astype(np.float16); ref["optimizer/vars/2"] = opt["2"][()]
    f.attrs["note"] = "attributes on the root group"
    for n in range(20):                               # enough attributes to push the object header into continuation blocks
        layers["conv2d"].attrs["a%d" % n] = np.arange(n + 1)
    layers["conv2d/vars/0"].attrs["unit"] = "none"
np.savez(sys.argv[2], **{k.replace("/", "|"): v for k, v in ref.items()})
with h5py.File(sys.argv[3] + "/latest.h5", "w", libver="latest") as f:
    f["x"] = np.zeros(3, np.float32)
with h5py.File(sys.argv[3] + "/chunked.h5", "w") as f:
    f.create_dataset("x", data=np.zeros((64, 64), np.float32), chunks=(8, 8), compression="gzip")
with h5py.File(sys.argv[3] + "/bigendian.h5", "w") as f:
    f.create_dataset("x", data=np.arange(4, dtype=">f4"))
"""


def test_the_reader_reads_what_real_h5py_writes_and_refuses_the_rest(tmp_path, h5py_version):
    run_h5py(WRITE_WITH_H5PY, str(tmp_path / "real.h5"), str(tmp_path / "ref.npz"), str(tmp_path))
    tree = H.read_file((tmp_path / "real.h5").read_bytes())
    z = np.load(tmp_path / "ref.npz")
    want = {k.replace("|", "/"): z[k] for k in z.files}
    got = dict(flat(tree))
    assert len(got) == 607 and set(got) == set(want) and all(same(got[k], want[k]) for k in want)
    assert len(tree["layers"]) == 302 and tree["layers"]["activation"]["vars"] == {} and tree["vars"] == {}
    assert got["optimizer/vars/0"].shape == () and got["optimizer/vars/2"].dtype == np.float16
    for name, what in (("latest", "superblock version"), ("chunked", "filtered|chunked"), ("bigendian", "big-endian")):
        with pytest.raises(H.Hdf5Unsupported, match=what):
            H.read_file((tmp_path / f"{name}.h5").read_bytes())


def test_a_keras_layout_archive_survives_real_h5py_in_the_middle(tmp_path, h5py_version):
    """model.weights.h5 out of a `.keras` archive written by keras_archive.save_keras, re-written by REAL h5py (read every dataset,
    create a fresh file), read back by keras_archive: the weights arrive bit for bit -- the archive is a file libhdf5 accepts and the
    loader accepts what libhdf5 writes."""
    import zipfile
    import torch
    from adunet_amd import keras_archive as K
    from adunet_amd.model import build_super_resolution_unet
    from tests.test_keras_archive_cpu import HostWeights
    model, _ = build_super_resolution_unet(0.5, depth_override=2, input_size=32, dtype=torch.float32)
    src = HostWeights(model, seed=5)
    K.save_keras(src, tmp_path / "m.keras")
    with zipfile.ZipFile(tmp_path / "m.keras") as z:
        (tmp_path / "inner.h5").write_bytes(z.read("model.weights.h5"))
    run_h5py("""
import sys, h5py
with h5py.File(sys.argv[1], "r") as a, h5py.File(sys.argv[2], "w") as b:
    def copy(name, obj):
        if isinstance(obj, h5py.Dataset): b.create_dataset(name, data=obj[()])
        else: b.require_group(name)
    a.visititems(copy)
""", str(tmp_path / "inner.h5"), str(tmp_path / "rewritten.weights.h5"))
    dst = HostWeights(model, seed=77)
    K.load_into(dst, tmp_path / "rewritten.weights.h5")
    assert set(dst.loaded) == set(src.w) and all(np.array_equal(dst.loaded[k], src.w[k]) for k in src.w)


def test_oracle_ssim_against_scikit_image(tmp_path, h5py_version):
    """The same second interpreter carries scikit-image 0.18: `structural_similarity(gaussian_weights=True, sigma=1.5,
    use_sample_covariance=False)` is Wang et al.'s SSIM with the 11-tap Gaussian window, K1 = 0.01, K2 = 0.03, averaged over the
    region the full window covers -- tf.image.ssim's definition, implemented by a third party.  oracle.metrics.ssim_per_image (what
    the device kernels are tested against) must agree with it on ordinary patches; TensorFlow itself stays unavailable."""
    from oracle import metrics as ref_metrics
    rng = np.random.default_rng(3)
    a = rng.random((3, 48, 40)).astype(np.float64)
    from scipy.ndimage import gaussian_filter
    a = np.stack([gaussian_filter(x, 1.5) for x in a])
    a = (a - a.min()) / (a.max() - a.min())
    b = np.clip(a + 0.08 * rng.standard_normal(a.shape), 0, 1)
    np.savez(tmp_path / "planes.npz", a=a, b=b)
    try:
        out = run_h5py("""
import sys, numpy as np, warnings
warnings.filterwarnings("ignore")
from skimage.metrics import structural_similarity
z = np.load(sys.argv[1])
print(" ".join(repr(float(structural_similarity(x, y, gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=1.0)))
               for x, y in zip(z["a"], z["b"])))
""", str(tmp_path / "planes.npz"))
    except AssertionError as exc:
        pytest.skip(f"scikit-image not usable in {CONDA_PY}: {exc}")
    theirs = np.array([float(v) for v in out.split()])
    ours = ref_metrics.ssim_per_image(a[..., None], b[..., None]).astype(np.float64)
    assert theirs.shape == (3,) and (theirs < 0.99).all() and (theirs > 0.2).all()          # ordinary patches, not degenerate ones
    assert np.abs(ours - theirs).max() < 2e-6, (ours, theirs)
