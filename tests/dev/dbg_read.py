"""TEST INFRASTRUCTURE (uses the oracle as the checker).  Dev tool (GPU box): one dumped read (tests/golden/sweep_reads_r05.npz) on the
device vs the oracle vs the reference's object code on the bench index of a workload, then the per-event trace (device single-slot
trace vs oracle) up to the first divergence.     python tests/dev/dbg_read.py <workload> <npz> <read>"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE  # noqa: E402
from uncalled_amd import capi  # noqa: E402

wl, npz, r = sys.argv[1], sys.argv[2], int(sys.argv[3])
raw = np.load(npz)[f"raw_{r}"]
prefix, codes, lens = bench.ensure_index(Path(os.environ.get("UNC_BENCH_CACHE", "/tmp/uncalled_amd_bench")), 0, lambda: None, wl, "cuda:0")
del codes
ix = capi.Index(prefix)
off = np.array([0, raw.size], dtype=np.uint64)
cal = capi.make_calib(1, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
m = capi.Mapper(ix, n_slots=64, n_waves=64)
h = m.map_batch(raw, off, cal)[0]
sig = po.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
oix = po.Index(prefix)
o = po.Mapper(oix).map_read(sig)
pyref.init(prefix)
rf = pyref.Mapper().map_read(sig)
for name, x in (("device", h), ("oracle", o)):
    print(name, {f: int(x[f]) for f in ("mapped", "event_i", "n_events", "n_nbr", "n_sa", "n_lf", "notes")})
print("reference", dict(mapped=rf.mapped, event_i=rf.event_i, n_events=rf.n_events, n_nbr=rf.n_nbr, n_sa=rf.n_sa, n_lf=rf.n_lf))
for mode, nm in ((pyref.SORT_PDQ_RESTATED, "pdq_restated"), (pyref.SORT_REVERSED_TIES, "reversed_ties")):
    pyref.set_sort_mode(mode)
    x = pyref.Mapper().map_read(sig)
    print("reference,", nm, dict(mapped=x.mapped, event_i=x.event_i, n_nbr=x.n_nbr, n_sa=x.n_sa, n_lf=x.n_lf))
pyref.set_sort_mode(pyref.SORT_STABLE)
# per-event: device trace vs oracle trace vs reference trace (path tables after every event)
mt = capi.Mapper(ix, n_slots=1)
om = po.Mapper(oix)
rm = pyref.Mapper()
steps = 0
for (dd, dp, dc, dmm, dls, dnl), (od, oe, op, oc, omm, ols, onl), (rd, re, rp, rc, rmm, rls, rnl) in zip(
        mt.trace(raw, cal[:1], max_clusters=1 << 17), om.trace(sig, max_clusters=1 << 17), rm.trace(sig, max_clusters=1 << 17)):
    ov, rv = op[op["length"] > 0], rp[rp["length"] > 0]
    why = None
    if len(dp) != len(ov) or len(ov) != len(rv):
        why = f"path count device {len(dp)} oracle {len(ov)} reference {len(rv)}  (buffer entries: oracle {len(op)} reference {len(rp)})"
    else:
        for f in ("fm_start", "fm_end", "kmer", "length", "event_moves", "seed_prob", "consec_stays", "sa_checked"):
            if not np.array_equal(dp[f], ov[f]) or not np.array_equal(ov[f], rv[f]):
                why = f"path field {f} differs (device==oracle: {np.array_equal(dp[f], ov[f])}, oracle==reference: {np.array_equal(ov[f], rv[f])})"
                break
    if why:
        print(f"first divergence after event {steps}: {why}")
        break
    steps += 1
    if len(op) >= 9990 or steps % 500 == 0:
        print("  event", steps, "valid paths", len(ov), "buffer entries oracle", len(op), "reference", len(rp), flush=True)
else:
    print(f"path tables identical through {steps} events (device == oracle == reference)")
hd = None
print("trace finish: oracle", {f: int(om.trace_finish()[f]) for f in ("event_i", "n_nbr")}, "reference n_nbr", rm.trace_finish().n_nbr)
