"""Structure pinned by the reference's own artefacts: the 15 committed Keras model.summary() dumps
(condensed into tests/golden/model_summaries.json by tests/golden/make_summary_fixture.py).
Both the oracle and the product graph must reproduce every layer row and every parameter count."""
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SUMMARIES = json.load(open(os.path.join(HERE, "golden", "model_summaries.json")))
TOTALS = {1: 520003, 2: 2144451, 3: 8637379, 4: 34599363, 5: 138427843}


def _cfg(s):
    m = re.match(r"U-Net_SR_scale([\d.]+)_depth(\d+)", s["model"])
    return float(m.group(1)), int(m.group(2))


def _norm(t):
    return "ClippedResidualAdd" if t == "ClipAdd" else t   # legacy alias, shared/custom_layers.py:142


@pytest.mark.parametrize("s", SUMMARIES, ids=[s["file"].split("/")[-1][:-18] + f"#{i}" for i, s in enumerate(SUMMARIES)])
def test_product_graph_matches_reference_summary(s):
    from adunet_amd.model import build_super_resolution_unet
    scale, depth = _cfg(s)
    model, info = build_super_resolution_unet(scale, depth_override=depth)
    assert model.name == s["model"]
    assert model.count_params() == s["total_params"] == TOTALS[depth]
    got = [(r.name, r.type, list(r.shape), r.params) for r in model.layers]
    want = [(l["name"], _norm(l["type"]), l["shape"], l["params"]) for l in s["layers"]]
    assert got == want
    assert info["depth"] == depth and info["base_channels"] == 64 and info["scale"] == scale


@pytest.mark.parametrize("s", SUMMARIES[::3], ids=lambda s: s["model"])
def test_oracle_graph_matches_reference_summary(s):
    from oracle.sr_unet import SRUNetOracle
    scale, depth = _cfg(s)
    m = SRUNetOracle(scale, depth, 256)
    assert m.name == s["model"] and m.count_params() == s["total_params"]
    got = [(r["name"], r["type"], list(r["shape"]), r["params"]) for r in m.layers]
    want = [(l["name"], _norm(l["type"]), l["shape"], l["params"]) for l in s["layers"]]
    assert got == want


def test_summary_text_and_flags():
    from adunet_amd.model import build_super_resolution_unet
    model, _ = build_super_resolution_unet(0.6, depth_override=4)
    lines = []
    model.summary(print_fn=lines.append)
    text = "\n".join(lines)
    assert 'Model: "U-Net_SR_scale0.60_depth4"' in text and "Total params: 34,599,363" in text
    assert model.sizes == [256, 154, 93, 56, 34]     # exp2_adaptive_depth_scale0.60 pyramid
    with pytest.raises(ValueError):
        build_super_resolution_unet(1.5)              # heuristic rejects scale outside (0.05, 1)
    with pytest.raises(ValueError):
        model.compile(loss="nonsense")
    with pytest.raises(NotImplementedError):
        model.compile(loss="combined")
