#!/usr/bin/env python3
"""Generates tests/golden/param_variant_goldens.npz from oracle/_ref (the reference's own sources compiled in place): the PAF
fields and work counters of the bundled example read + the first 8 synthetic reads of ref_goldens.npz under every parameter set
of tests/parity_cases.py PARAM_VARIANTS (Mapper::PRMS set through ref_set_params before the Mapper is constructed).
Container-only (needs /root/reference via `make -C oracle ref`); the output is committed so that the oracle can be checked on
these parameter sets where the reference does not exist."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as po  # noqa: E402  (only for the Params structure's defaults)
from oracle import pyref  # noqa: E402
from tests.golden.make_goldens import PREFIX, hit_tuple  # noqa: E402
from tests.parity_cases import PARAM_VARIANTS, variant_params  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE  # noqa: E402

G = Path(__file__).resolve().parent


def main():
    gold = np.load(G / "ref_goldens.npz")
    pyref.init(PREFIX)
    off = gold["sim_offsets"]
    sigs = [gold["ex_calibrated"]] + [pyref.calibrate(gold["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
                                      for i in range(8)]
    out = []
    for ov in PARAM_VARIANTS:
        pyref.set_params(variant_params(po.default_params(), ov))
        m = pyref.Mapper()
        out.append(np.stack([hit_tuple(m.map_read(s)) for s in sigs]))
    pyref.set_params(po.default_params())
    np.savez_compressed(G / "param_variant_goldens.npz", hits=np.stack(out), variants=np.array([repr(sorted(v.items())) for v in PARAM_VARIANTS]))
    print("wrote", G / "param_variant_goldens.npz", np.stack(out).shape, "mapped per variant", np.stack(out)[:, :, 0].sum(axis=1))


if __name__ == "__main__":
    main()
