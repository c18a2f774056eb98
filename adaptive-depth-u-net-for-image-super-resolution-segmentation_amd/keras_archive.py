"""Keras-3 `.keras` archives and `.weights.h5` files without h5py (checkpoint interchange).

What the reference writes and reads (TensorFlow 2.16.1 / Keras 3.3.3, Super_resolution/requirement.txt:4,8):
  * `ModelCheckpoint(... .keras)` / `model.save` (Super_resolution/code/train_adaptive_unet.py:531,617;
    Segmenation/code/unet_vinillia.py:276,292): a zip of `metadata.json`, `config.json`, `model.weights.h5`;
  * `model.load_weights(resume)` (train_adaptive_unet.py:511-516) and the fallback of `load_checkpoint_model`
    (evaluate_model.py:71-91: rebuild the architecture, then `load_weights`) read ONLY `model.weights.h5` out of it.

STATUS.  The HDF5 container (hdf5_min.py) is pinned both ways against libhdf5 1.10.6 through the h5py of the image's second
interpreter (tests/test_against_second_interpreter.py: incl. a model.weights.h5 from `save_keras` re-written by real h5py and loaded
back bit for bit).  The STORE NAMING below is restated from Keras 3's `saving_lib` and is NOT pinned: Keras is absent from both
interpreters and the reference git-ignores its `.keras` checkpoints, so no archive written by Keras exists to read.
  model.weights.h5:  /layers/<store name>/vars/<i>   float32, i = position in trainable + non-trainable variables
                     /vars                           (empty for a functional model)
                     /optimizer/vars/<i>             (written by Keras when the model is compiled; ignored on import)
  <store name> is NOT the layer's name: `_save_container_state` numbers the layers by class in `model.layers` order --
  to_snake_case(class name), then `_1`, `_2`, ... (a Conv2D named "residual_rgb" is stored as e.g. `conv2d_9`), every layer
  gets a group, layers without variables an empty `vars`.
  Variables per class: Conv2D / Conv2DTranspose [kernel, bias]; LayerNormalization [gamma, beta];
  BatchNormalization [gamma, beta, moving_mean, moving_variance].
`config.json` is generated best-effort (functional config with the reference's registered custom-object names
`resize>ResizeByScale`, `resize>ResizeToMatch`, `utils>ClippedResidualAdd`, shared/custom_layers.py:85,114,134); the
reference's own loader does not depend on it deserialising (evaluate_model.py:79-91 falls back to rebuild + load_weights).
The HDF5 container itself: hdf5_min.py (same status).
"""
from __future__ import annotations

import json
import re
import zipfile
from datetime import datetime
from typing import Dict, List, Tuple

import numpy as np

from . import hdf5_min

KERAS_VERSION = "3.3.3"                      # Super_resolution/requirement.txt:8
LAYER_VARIABLES = {"Conv2D": ("kernel", "bias"), "Conv2DTranspose": ("kernel", "bias"), "LayerNormalization": ("gamma", "beta"),
                   "BatchNormalization": ("gamma", "beta", "moving_mean", "moving_variance")}
CUSTOM_OBJECTS = {"ResizeByScale": "resize>ResizeByScale", "ResizeToMatch": "resize>ResizeToMatch",
                  "ClippedResidualAdd": "utils>ClippedResidualAdd"}


def to_snake_case(name: str) -> str:
    """keras.src.utils.naming.to_snake_case, regex for regex."""
    name = re.sub(r"\W+", "", name)
    name = re.sub("(.)([A-Z][a-z]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", name).lower()


def store_layout(model) -> List[Tuple[str, List[str]]]:
    """[(store name, [this model's variable names in Keras' vars order])] for every layer of `model.layers`, in order."""
    used: Dict[str, int] = {}
    out = []
    for row in model.layers:
        base = to_snake_case(row.type)
        if base in used:
            used[base] += 1
            store = f"{base}_{used[base]}"
        else:
            used[base] = 0
            store = base
        out.append((store, [f"{row.name}/{v}" for v in LAYER_VARIABLES.get(row.type, ())]))
    return out


def weights_tree(model) -> dict:
    weights = model.get_weights()
    layers = {}
    for store, names in store_layout(model):
        missing = [n for n in names if n not in weights]
        if missing:
            raise KeyError(f"model has no variable {missing[0]} for store entry {store}")
        layers[store] = {"vars": {str(i): np.asarray(weights[n], np.float32) for i, n in enumerate(names)}}
    return {"layers": layers, "vars": {}}


def weights_from_tree(model, tree: dict) -> Dict[str, np.ndarray]:
    """This model's {variable name: array} from a parsed model.weights.h5, matched by store position and checked by shape."""
    if "layers" not in tree:
        hint = " (a Keras-2 / legacy `.h5` with `layer_names` attributes is a different format)" if tree else ""
        raise ValueError(f"no /layers group in the weights file{hint}")
    shapes = {n: tuple(s) for n, (_, s) in model.index.items()}
    shapes.update({n: tuple(s) for n, (_, s) in getattr(model, "state_index", {}).items()})
    out: Dict[str, np.ndarray] = {}
    for store, names in store_layout(model):
        if not names:
            continue
        group = tree["layers"].get(store)
        if group is None:
            raise ValueError(f"weights file has no /layers/{store} (the archive belongs to another architecture)")
        vs = group.get("vars", {})
        if len(vs) != len(names):
            raise ValueError(f"/layers/{store}/vars holds {len(vs)} variables, the model's layer has {len(names)}")
        for i, n in enumerate(names):
            arr = np.asarray(vs[str(i)])
            if tuple(arr.shape) != shapes[n]:
                raise ValueError(f"/layers/{store}/vars/{i}: shape {tuple(arr.shape)}, the model's {n} is {shapes[n]}")
            out[n] = arr.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------ config.json (best effort)
def _layer_config(model, row) -> dict:
    cfg = {"name": row.name, "trainable": True, "dtype": "float32"}
    cs = getattr(model, "convs", {}).get(row.name)
    if row.type == "Conv2D" and cs is not None:
        head = row.name in ("residual_rgb", "lesion_mask", "mask_logits")
        act = "relu" if (cs.ln is None and not head) else ("sigmoid" if row.name in ("lesion_mask", "mask_logits") else "linear")
        cfg.update(filters=cs.cout, kernel_size=[cs.k, cs.k], strides=[1, 1], padding="same", activation=act, use_bias=True)
    elif row.type == "Conv2DTranspose":
        cfg.update(filters=row.shape[-1], kernel_size=[2, 2], strides=[2, 2], padding="same", activation="linear", use_bias=True)
    elif row.type in ("LayerNormalization", "BatchNormalization"):
        cfg.update(axis=[-1] if row.type == "LayerNormalization" else -1, epsilon=1e-3, center=True, scale=True)
        if row.type == "BatchNormalization":
            cfg.update(momentum=0.99)
    elif row.type == "Activation":
        cfg.update(activation="relu")
    elif row.type == "MaxPooling2D":
        cfg.update(pool_size=[2, 2], strides=[2, 2], padding="valid")
    elif row.type == "UpSampling2D":
        cfg.update(size=[2, 2], interpolation="bilinear")
    elif row.type == "Concatenate":
        cfg.update(axis=-1)
    elif row.type == "ResizeByScale":
        cfg.update(scale=float(getattr(model, "scale", 0.0)), method="bilinear", antialias=True)
    elif row.type == "ResizeToMatch":
        cfg.update(method="bilinear", antialias=True)
    elif row.type == "InputLayer":
        cfg = {"batch_shape": [None, *row.shape], "dtype": "float32", "sparse": False, "name": row.name}
    return cfg


SHARED_ARITY = {"ResizeByScale": 1, "ResizeToMatch": 2}       # the ONE enc_down / dec_up instance is called once per level (:233-234)


def model_config(model) -> dict:
    """Functional config in Keras 3's shape.  A shared layer (enc_down, dec_up) has one inbound node per call; the layer rows
    list the inputs of all its calls in call order, and in these graphs every call has exactly one consumer, met in the same
    order in `model.layers` -- which gives each consumer its node index and the call's output shape (the consumer's own spatial
    size, the resized tensor's channels)."""
    rows = {r.name: r for r in model.layers}
    shapes = {r.name: [None, *r.shape] for r in model.layers}
    shared = {r.name: SHARED_ARITY[r.type] for r in model.layers if r.type in SHARED_ARITY and len(r.inbound) > SHARED_ARITY[r.type]}
    next_call = {name: 0 for name in shared}

    def tensor(src, node=0, shape=None):
        return {"class_name": "__keras_tensor__",
                "config": {"shape": shape or shapes[src], "dtype": "float32", "keras_history": [src, node, 0]}}

    def input_of(consumer, src):
        if src not in shared:
            return tensor(src)
        k = next_call[src]
        next_call[src] += 1
        resized = rows[src].inbound[k * shared[src]]                       # the tensor that call k resizes keeps its channels
        return tensor(src, k, [None, consumer.shape[0], consumer.shape[1], shapes[resized][-1]])

    layers = []
    for row in model.layers:
        custom = CUSTOM_OBJECTS.get(row.type)
        if row.name in shared:                                             # its own inputs never come from a shared layer's later call
            ar = shared[row.name]
            calls = [[tensor(s) if s not in shared else input_of(row, s) for s in row.inbound[i:i + ar]]
                     for i in range(0, len(row.inbound), ar)]
        else:
            calls = [[input_of(row, s) for s in row.inbound]] if row.inbound else []
        inbound = [{"args": [args] if len(args) > 1 else args, "kwargs": {}} for args in calls]
        layers.append({"module": "shared.custom_layers" if custom else "keras.layers", "class_name": row.type,
                       "config": _layer_config(model, row), "registered_name": custom, "name": row.name, "inbound_nodes": inbound})
    return {"module": "keras.src.models.functional", "class_name": "Functional",
            "config": {"name": model.name, "trainable": True, "layers": layers,
                       "input_layers": [[model.layers[0].name, 0, 0]], "output_layers": [[model.layers[-1].name, 0, 0]]},
            "registered_name": "Functional", "build_config": {"input_shape": None}, "compile_config": None}


# ------------------------------------------------------------------------------------------------------------ files
def save_weights_h5(model, path) -> None:
    with open(path, "wb") as fh:
        fh.write(hdf5_min.write_file(weights_tree(model)))


def save_keras(model, path) -> None:
    """`model.save("x.keras")`: metadata.json + config.json + model.weights.h5 in one (stored, uncompressed) zip."""
    meta = {"keras_version": KERAS_VERSION, "date_saved": datetime.now().strftime("%Y-%m-%d@%H:%M:%S")}
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("metadata.json", json.dumps(meta))
        z.writestr("config.json", json.dumps(model_config(model)))
        z.writestr("model.weights.h5", hdf5_min.write_file(weights_tree(model)))


def read_weights(path) -> dict:
    """The parsed weights tree of a `.keras` archive or a bare `.weights.h5` / `.h5` file."""
    path = str(path)
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            if "model.weights.h5" not in z.namelist():
                raise ValueError(f"{path}: a zip without model.weights.h5 is not a Keras-3 archive")
            data = z.read("model.weights.h5")
    else:
        with open(path, "rb") as fh:
            data = fh.read()
    return hdf5_min.read_file(data)


def load_into(model, path) -> None:
    """`model.load_weights(path)` for `.keras` / `.weights.h5`: weights (and BatchNorm moving statistics) by store position."""
    model.set_weights(weights_from_tree(model, read_weights(path)))


def describe(path) -> dict:
    """metadata.json and the model name / layer count of config.json of a `.keras` archive (for messages and tests)."""
    with zipfile.ZipFile(str(path)) as z:
        meta = json.loads(z.read("metadata.json"))
        cfg = json.loads(z.read("config.json"))
    return {"metadata": meta, "model_name": cfg["config"]["name"], "layers": len(cfg["config"]["layers"]),
            "custom": sorted({l["registered_name"] for l in cfg["config"]["layers"] if l["registered_name"]})}
