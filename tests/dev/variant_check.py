"""TEST INFRASTRUCTURE (uses the oracle as the checker).  Dev tool: run the parity cases against an explicitly named build of the HIP library (codegen experiments)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from oracle import pyoracle as po
from uncalled_amd import capi
from tests.helpers import HIT_INT_FIELDS, oracle_hits
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE

lib = capi.load(sys.argv[1])
g = ROOT / "tests/golden"
gold = np.load(g / "ref_goldens.npz")
prefix = g / "example_index/example_ref"
ix = capi.Index(prefix, lib=lib)
oix = po.Index(prefix)
n = 48
off = gold["sim_offsets"][:n + 1].copy()
raw = gold["sim_signal"][:int(off[n])]
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
want = oracle_hits(oix, raw, off, cal)
for rep in range(3):
    m = capi.Mapper(ix, n_slots=0 if rep else 16)
    hits = m.map_batch(raw, off, cal, allow_overflow=True)
    bad = [(i, f, int(hits[i][f]), int(want[i][f])) for i in range(n) for f in HIT_INT_FIELDS if int(hits[i][f]) != int(want[i][f])]
    print(f"rep {rep}: mismatching reads {len(set(b[0] for b in bad))}/{n}", bad[:3], flush=True)
    m.close()
print("DONE", flush=True)
