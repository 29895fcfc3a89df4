"""TEST INFRASTRUCTURE (uses the oracle as the checker).  Dev tool (GPU box): who is right at GRCh38 scale?  Device batch vs oracle restatement vs the reference's own object code
on the same reads, then a per-event trace (device single-slot trace API vs oracle) of the first read that differs."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import bench  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE  # noqa: E402
from tools.simulate_reads_torch import simulate_reads_torch  # noqa: E402
from uncalled_amd import capi  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "grch38"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cache = Path(os.environ.get("UNC_BENCH_CACHE", "/tmp/uncalled_amd_bench"))
prefix, codes, lens = bench.ensure_index(cache, 0, lambda: None, wl, "cuda:0")
sim = simulate_reads_torch(codes, lens, n, seed=44, device="cuda:0")
raw = sim["signal"].cpu().numpy()
off = sim["offsets"]
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
ix = capi.Index(prefix)
print("thresholds", ix.thresholds()[:16], "uncl", open(str(prefix) + ".uncl").read()[:300])
m = capi.Mapper(ix)
print("geometry", m.geometry())
hits = m.map_batch(raw, off, cal)
m1 = capi.Mapper(ix, n_slots=64, n_waves=64)      # plain kernel, one slot per wavefront
hits1 = m1.map_batch(raw, off, cal, allow_overflow=True)
sig = po.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
oix = po.Index(prefix)
t0 = time.time()
want, _ = po.map_batch(oix, sig, off, min(n, 128))
print("oracle", time.time() - t0, "s")
ref = None
if pyref.available():
    pyref.init(prefix)
    t0 = time.time()
    ref, _ = pyref.map_batch(sig, off, min(n, 128))
    print("reference", time.time() - t0, "s")
bad = []
for i in range(n):
    r = (ref[i].event_i, ref[i].n_nbr, ref[i].mapped) if ref is not None else None
    line = (i, "dev", int(hits["event_i"][i]), int(hits["n_nbr"][i]), int(hits["mapped"][i]), int(hits["status"][i]),
            "plain", int(hits1["event_i"][i]), int(hits1["n_nbr"][i]), int(hits1["status"][i]),
            "ora", int(want["event_i"][i]), int(want["n_nbr"][i]), int(want["mapped"][i]), "ref", r)
    ok = int(hits["event_i"][i]) == int(want["event_i"][i]) and int(hits["n_nbr"][i]) == int(want["n_nbr"][i])
    if not ok:
        bad.append(i)
    print(*line, "" if ok else "  <<<< MISMATCH")
print("mismatching reads:", bad)
if bad:
    i = bad[0]
    r = raw[int(off[i]):int(off[i + 1])]
    mt = capi.Mapper(ix, n_slots=1)
    om = po.Mapper(oix)
    steps = 0
    for (dd, dp, dc, dmm, dls, dnl), (od, oe, op, oc, omm, ols, onl) in zip(mt.trace(r, cal[:1], max_clusters=1 << 17), om.trace(sig[int(off[i]):int(off[i + 1])], max_clusters=1 << 17)):
        ov = op[op["length"] > 0]
        why = None
        if len(dp) != len(ov):
            why = f"path count {len(dp)} vs {len(ov)}"
        else:
            for f in ("fm_start", "fm_end", "kmer", "length", "event_moves", "seed_prob", "consec_stays", "sa_checked"):
                if not np.array_equal(dp[f], ov[f]):
                    j = int(np.flatnonzero(dp[f] != ov[f])[0])
                    why = f"path field {f} at {j}: {dp[j]} vs {ov[j]}"
                    break
        if why is None and not np.array_equal(dc, oc):
            if len(dc) != len(oc):
                why = f"cluster count {len(dc)} vs {len(oc)}"
            else:
                j = int(np.flatnonzero(dc != oc)[0])
                why = f"cluster {j}: {dc[j]} vs {oc[j]}  prev {dc[j - 1] if j else None} next {dc[j + 1] if j + 1 < len(dc) else None}"
        if why is None and (dls != ols or dnl != onl):
            why = f"len_sum {dls} vs {ols}, n_lens {dnl} vs {onl}"
        if why is None and dd != od:
            why = f"done {dd} vs {od}; max_map {dmm} vs {omm}"
        if why:
            print(f"read {i}: first divergence after event {steps}: {why}")
            print("  n_paths", len(dp), "n_clusters", len(dc), len(oc), "max_map", dmm, omm)
            break
        steps += 1
        if steps % 200 == 0:
            print("  trace ok through event", steps, "paths", len(dp), "clusters", len(dc), flush=True)
    else:
        print(f"read {i}: trace identical through {steps} events (single slot, plain kernel)")
