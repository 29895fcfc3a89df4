"""Dev tool (GPU box): which PHASE's cycles grow when a GRCh38 launch lands on its slow level?  (round-5 review, item 2)

    python tools/dev/grch38_phase_spread.py [n_reads = 250000] [launches = 6]

Maps the bench's own GRCh38 batch `launches` times on ONE mapper with the cycle-counting instantiation of k_map (2 % slower than the
plain one) and prints, per launch: k_map ms, wavefront busy share, pool high-water, and the twelve phase-cycle counters summed over the
batch's reads (absolute, in units of 10^9 shader cycles) -- then for the slowest against the fastest launch the counters' ratio."""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
workload = os.environ.get("SPREAD_WORKLOAD", "grch38")
pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
# SPREAD_LIB: a variant library (tools/dev/build_variants.py); with dbgseed="-DUNC_DBG_SEED=1" counters 8..11 are add_seeds' own
# (rounds, nodes walked, cycles inside the pool's ring, longest walk of one seed) and phase E's parts sit in counter 1
LIB = capi.load(os.environ["SPREAD_LIB"]) if os.environ.get("SPREAD_LIB") else capi.load()
DBG = "dbgseed" in os.environ.get("SPREAD_LIB", "")
ix = capi.Index(pre, lib=LIB)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
# argv[3]: mapper variants, comma separated: "auto" (the default pool), "pool=<chunks>", "slice=<events>"
variants = (sys.argv[3] if len(sys.argv) > 3 else "auto").split(",")
NAMES = ("probs", "extend_rest", "sort", "walk", "sources", "sa", "add_seed", "rest", "e1_parents", "e2_fm", "e3_slots", "e4_children")


def deciles(x, by, what):
    """sum of x per decile of `by` (reads ordered by `by`)"""
    o = np.argsort(by, kind="stable")
    return [float(x[c].sum()) for c in np.array_split(o, 10)]


for var in variants:
    kw = {}
    if var.startswith("pool="):
        kw["pool_chunks"] = int(var[5:])
    if var.startswith("slice="):
        kw["slice_events"] = int(var[6:])
    m = capi.Mapper(ix, **kw)
    m.set_profile(True)
    rows, first = [], None
    for i in range(launches):
        hits = m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
        ms = m.last_timing()[1]
        pc = m.last_phase_cycles()
        u = m.pool_usage()
        d = capi.hits_digest(hits)
        first = first or d
        per = m.last_read_cycles(n)
        rows.append((ms, pc, u["high_water_last_batch"], m.last_wave_busy(), per))
        xcc = (per[:, 13] & 0xF).astype(np.int64)
        seed_by_xcc = np.bincount(xcc, weights=per[:, 6].astype(np.float64), minlength=8)[:8]
        reads_by_xcc = np.bincount(xcc, minlength=8)[:8]
        print(f"[{var}] launch {i}: k_map {ms:8.1f} ms  wave_busy {m.last_wave_busy():.3f}  pool {u['chunks']} chunks, high-water {u['high_water_last_batch']}, resizes {u['resizes']}  "
              f"remap {m.last_remap()[0]}  hits {'same' if d == first else 'DIFFER'}  "
              + " ".join(f"{k}={v / 1e9:.1f}" for k, v in pc.items()), flush=True)
        rot = ((per[:, 13] & 0xF).astype(np.int64) - ((per[:, 13] >> 24) & 0xFF).astype(np.int64)) % 8
        cu = ((per[:, 13] >> 16) & 0xF).astype(np.int64)      # HW_ID bits 11:8
        print(f"[{var}]    (XCC_ID - workgroup) mod 8 over the reads: " + " ".join(str(int(c)) for c in np.bincount(rot, minlength=8)[:8]), flush=True)
        print(f"[{var}]    add_seed cycles per read decided on XCC 0..7 (10^3): " + " ".join(f"{a / max(1, b) / 1e3:.0f}" for a, b in zip(seed_by_xcc, reads_by_xcc))
              + "   reads decided there: " + " ".join(str(int(b)) for b in reads_by_xcc), flush=True)
        # the XCDs that decide few reads: which reads are those?
        few = [x for x in range(8) if reads_by_xcc[x] * 4 < reads_by_xcc.max()]
        tot = per[:, :12].astype(np.float64).sum(axis=1)
        for x in few:
            idx = np.flatnonzero(xcc == x)
            print(f"[{var}]    XCC {x}: {idx.size} reads decided; read numbers min {idx.min()} p25 {int(np.percentile(idx, 25))} median {int(np.median(idx))} p75 {int(np.percentile(idx, 75))} max {idx.max()}; "
                  f"events per read median {int(np.median(hits['n_events'][idx]))}, event_i median {int(np.median(hits['event_i'][idx]))}; mapped {float(hits['mapped'][idx].mean()):.2f}; "
                  f"add_seed / all cycles of these reads {per[idx, 6].sum() / max(1.0, tot[idx].sum()):.3f}; residence ticks median {np.median(per[idx, 12]):.3g} (launch: {ms * 1e-3 * 1e8:.3g} at 100 MHz)", flush=True)
        if DBG:
            for name, sel in (("reads decided on the XCDs that decide few", np.isin(xcc, few)), ("all other reads", ~np.isin(xcc, few))):
                q = per[sel]
                if q.shape[0]:
                    print(f"[{var}]    {name}: {q.shape[0]} reads; per read: add_seed cycles {q[:, 6].mean():.3g}, rounds {q[:, 8].mean():.3g}, nodes walked {q[:, 9].mean():.3g}, "
                          f"cycles inside the pool's ring {q[:, 10].mean():.3g}, longest walk of one seed: median {np.median(q[:, 11]):.0f} max {q[:, 11].max()}; "
                          f"cycles per node walked {q[:, 6].sum() / max(1, q[:, 9].sum()):.0f}", flush=True)
        top = np.argsort(per[:, 6])[::-1][:12]
        print(f"[{var}]    top reads by add_seed cycles: " + "; ".join(f"#{int(i)} xcc {int(xcc[i])} cu {int(cu[i])} seed {per[i, 6] / 1e9:.1f}G all {tot[i] / 1e9:.1f}G ev {int(hits['event_i'][i])} nsa {int(hits['n_sa'][i])} m {int(hits['mapped'][i])}" + (f" rounds {int(per[i, 8])} nodes {int(per[i, 9])} ring {per[i, 10] / 1e9:.2f}G longest {int(per[i, 11])}" if DBG else "") for i in top), flush=True)
        print(f"[{var}]    all cycles of the reads decided per XCC (10^12): " + " ".join(f"{np.bincount(xcc, weights=tot, minlength=8)[x] / 1e12:.1f}" for x in range(8)), flush=True)
    if len(rows) > 1:
        slow, fast = max(rows, key=lambda r: r[0]), min(rows, key=lambda r: r[0])
        print(f"[{var}] slowest {slow[0]:.0f} ms / fastest {fast[0]:.0f} ms = {slow[0] / fast[0]:.3f}; per phase, cycles of the slowest / the fastest launch:")
        tot_s, tot_f = sum(slow[1].values()), sum(fast[1].values())
        print("  " + " ".join(f"{k}={slow[1][k] / fast[1][k]:.3f}" for k in slow[1] if fast[1][k]) + f"  all={tot_s / tot_f:.3f}")
        if tot_s != tot_f:
            print("  share of the DIFFERENCE in cycles per phase: " + " ".join(f"{k}={(slow[1][k] - fast[1][k]) / (tot_s - tot_f):.2f}" for k in slow[1]))
        # the same reads in the two launches: does every read pay, or the deep ones?  Reads in deciles of their add_seed cycles in the FAST launch
        a_s, a_f = slow[4][:, 6].astype(np.float64), fast[4][:, 6].astype(np.float64)
        ds, df = deciles(a_s, a_f, "add_seed"), deciles(a_f, a_f, "add_seed")
        print("  add_seed cycles slow / fast per decile of reads (by the read's add_seed cycles in the fast launch, shallow -> deep): "
              + " ".join(f"{x / y:.2f}" if y else "-" for x, y in zip(ds, df)))
        print("  share of all add_seed cycles in the decile (fast launch): " + " ".join(f"{y / max(1.0, a_f.sum()):.3f}" for y in df))
        r = a_s[a_f > 0] / a_f[a_f > 0]
        print("  per-read ratio add_seed slow / fast: p5 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f" % tuple(np.percentile(r, [5, 25, 50, 75, 95])))
        t_s, t_f = slow[4][:, 12].astype(np.float64), fast[4][:, 12].astype(np.float64)
        print("  residence ticks slow / fast: sum %.3f; per decile of residence in the fast launch: " % (t_s.sum() / t_f.sum())
              + " ".join(f"{x / y:.2f}" if y else "-" for x, y in zip(deciles(t_s, t_f, ""), deciles(t_f, t_f, ""))))
        other_s = slow[4][:, :12].astype(np.float64).sum(axis=1) - a_s
        other_f = fast[4][:, :12].astype(np.float64).sum(axis=1) - a_f
        print("  all OTHER phases' cycles slow / fast per the same deciles: " + " ".join(f"{x / y:.2f}" if y else "-" for x, y in zip(deciles(other_s, a_f, ""), deciles(other_f, a_f, ""))))
    m.close()
