"""Condense the reference's 15 Keras ``model.summary()`` dumps into one JSON fixture.

Run in the build container (reads /root/reference as *data*; nothing is imported or executed):
    python tests/golden/make_summary_fixture.py
Output: tests/golden/model_summaries.json -- per dump: model name, ordered rows
(layer name, layer type, output shape, param count) and the "Total params" figure.
Layer names that Keras truncated with an ellipsis are re-derived from their position
(layer_normalization_k counts up in order), which is how Keras numbers them.
"""
import glob
import json
import os
import re

ROOT = "/root/reference/Super_resolution/experiments"
OUT = os.path.join(os.path.dirname(__file__), "model_summaries.json")


def parse(path):
    rows, cur = [], None
    name, total = None, None
    for line in open(path, encoding="utf-8"):
        m = re.match(r'Model: "(.*)"', line)
        if m:
            name = m.group(1)
        m = re.search(r"Total params: ([\d,]+)", line)
        if m:
            total = int(m.group(1).replace(",", ""))
        if line.startswith("│"):
            cells = [c.strip() for c in line.strip().strip("│").split("│")]
            if cur is None:
                cur = ["", "", "", ""]
            for i in range(4):
                cur[i] += (" " if cur[i] and cells[i] else "") + cells[i]
        elif line.startswith(("├", "└")) and cur is not None:
            rows.append(cur)
            cur = None
    out, ln_idx = [], 0
    for lay, shape, params, _ in rows:
        m = re.match(r"(\S+)\s*\((.*)\)?$", lay.replace(" ", "", 0))
        lname, ltype = lay.split("(")[0].strip(), lay.split("(")[1].strip(") ")
        if ltype.startswith("LayerNormalizatio"):
            ltype = "LayerNormalization"
            lname = "layer_normalization" if ln_idx == 0 else f"layer_normalization_{ln_idx}"
            ln_idx += 1
        dims = [int(v) for v in re.findall(r"\d+", shape)]
        out.append({"name": lname, "type": ltype, "shape": dims, "params": int(params.replace(",", ""))})
    return {"file": os.path.relpath(path, "/root/reference"), "model": name, "total_params": total, "layers": out}


if __name__ == "__main__":
    files = sorted(glob.glob(os.path.join(ROOT, "*", "model_summary", "*.txt")))
    data = [parse(f) for f in files]
    json.dump(data, open(OUT, "w"), indent=0, separators=(",", ":"))
    print(len(data), "summaries ->", OUT)
