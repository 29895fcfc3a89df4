#!/usr/bin/env python3
"""Golden `.uncl` threshold lines from the reference's own `uncalled index` logic: its Python IndexParameterizer
(uncalled/index.py:53-209, executed in place from /root/reference) fed by its own C++ self_align
(src/self_align_ref.cpp:34-91, compiled in place into oracle/_ref).  Container-only; output committed as
tests/golden/uncl_goldens.json together with the self_align trajectories' digest."""
import argparse
import hashlib
import json
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyref  # noqa: E402
from uncalled_amd.build_index import build_from_codes, synthetic_genome  # noqa: E402

REF_INDEX_PY = Path("/root/reference/uncalled/index.py")


def reference_uncl(prefix):
    src = REF_INDEX_PY.read_text().replace("import uncalled as unc", "unc = _UNC_STUB")
    stub = types.SimpleNamespace(self_align=lambda p, d: [list(map(int, x)) for x in pyref.self_align(p, d)])
    ns = {"__file__": str(REF_INDEX_PY), "__name__": "ref_index", "_UNC_STUB": stub}
    exec(compile(src, str(REF_INDEX_PY), "exec"), ns)
    # defaults of `uncalled index` (uncalled/args.py:86-140)
    args = argparse.Namespace(bwa_prefix=str(prefix), matchpr1=0.6334, matchpr2=0.9838, max_sample_dist=100,
                              min_samples=50000, max_samples=1000000, kmer_len=5, pathlen_percentile=0.05, max_replen=100)
    p = ns["IndexParameterizer"](args)
    with tempfile.TemporaryDirectory() as d:
        p.out_fname = str(Path(d) / "x.uncl")
        p.add_preset("default", tgt_speed=115)       # scripts/uncalled:58
        p.add_preset("speed_60", tgt_speed=60)
        p.write()
        return Path(p.out_fname).read_text()


def digest(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(np.asarray(p, dtype=np.uint64).tobytes())
        h.update(b"|")
    return h.hexdigest()


def main():
    g = Path(__file__).resolve().parent
    out = {}
    ex = g / "example_index" / "example_ref"
    out["example"] = {"uncl": reference_uncl(ex), "bundled": (g / "example_index" / "example_ref.uncl").read_text(),
                      "self_align_dist1_sha256": digest(pyref.self_align(ex, 1))}
    with tempfile.TemporaryDirectory() as d:
        names, lens, codes = synthetic_genome(3, 600000, seed=5)
        prefix = Path(d) / "syn600k"
        build_from_codes(prefix, names, [""] * 3, lens, codes)
        out["syn600k_seed5_3contigs"] = {"uncl": reference_uncl(prefix), "self_align_dist12_sha256": digest(pyref.self_align(prefix, 12))}
    (g / "uncl_goldens.json").write_text(json.dumps(out, indent=1))
    for k, v in out.items():
        print(k, v["uncl"].strip().replace("\n", " || "))
    print("bundled:", out["example"]["bundled"].strip())


if __name__ == "__main__":
    main()
