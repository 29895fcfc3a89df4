#!/usr/bin/env python3
"""Generates uncalled_amd/data/r94_5mers_threshs.npz, the (threshold, fraction of events matching, mean k-mers
matched) table `uncalled index` interpolates in (uncalled/index.py:118-134 reads uncalled/conf/r94_5mers_threshs.txt).
Same role as tools/gen_model_table.py: container-only (reads /root/reference), the output is committed so that a
fresh clone can run `python -m uncalled_amd index` and bench.py without the reference tree."""
import sys
from pathlib import Path

import numpy as np

SRC = Path("/root/reference/uncalled/conf/r94_5mers_threshs.txt")
OUT = Path(__file__).resolve().parents[1] / "uncalled_amd" / "data" / "r94_5mers_threshs.npz"


def main():
    rows = np.array([[float(x) for x in l.split()] for l in SRC.read_text().splitlines() if l.strip()], dtype=np.float64)
    assert rows.shape == (4901, 3), rows.shape
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, thresh=rows[:, 0], freq=rows[:, 1], count=rows[:, 2])
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
