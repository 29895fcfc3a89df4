"""Bench-scale parity sweep (GPU box; under tests/ because it uses the oracle as the checker -- test infrastructure, never timed).

    python tests/dev/parity_sweep.py <workload: ecoli | chr20 | grch38> [n_checked = 10240] [threads = 64]

Maps the BENCH's own batch of that workload on the GPU (50 000 / 200 000 / 250 000 reads, same seed as bench.py), draws n_checked
reads at random ACROSS the batch (seeded) and maps those with the reference's own object code (oracle/_ref, stable tie order) on the
host threads (a read that disagrees is mapped again by a fresh reference Mapper: the threads' Mappers carry sources_added_ from read to read); compares EVERY field the reference's harness reports -- mapped, strand, the three read and three reference coordinates,
reference name, matches, event counts, mean event length (bit pattern) and the three work counters -- and writes one JSON line.
Round-4 review, item 4: the bench checks 1 024 GRCh38 reads per run (0.4 %); every scale-only defect so far lived on GRCh38."""
import json
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from oracle import pyoracle as po
from oracle import pyref
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

workload = sys.argv[1]
if workload.startswith("rt:"):
    # the CHUNKED path at scale: python tests/dev/parity_sweep.py rt:<ecoli | chr20> [reads per channel = 12] [rounds = 90]
    # 512 channels x reads_per_channel reads through unc_rt_* (bench.py's realtime workload, UNC_RT_TEAM wavefronts per channel), then
    # EVERY read every channel finished before its list wrapped against the reference's own chunk path (Mapper::new_read(Chunk&) /
    # add_chunk / process_chunk / map_chunk, one Mapper per channel, same per-channel order): PAF columns.  Round-5 review: the chunked
    # path -- other event-detector and normaliser code than the batch path -- had been checked on about a hundred reads per run.
    import argparse
    ref = workload.partition(":")[2] or "ecoli"
    rpc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 90
    t0 = time.time()
    pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, ref, "cuda:0")
    ix = capi.Index(pre)
    a = argparse.Namespace(channels=512)
    out = bench.realtime_workload(a, ix, pre, codes, lens, 0, ref, rounds, 0, verify_channels=512, cpu_budget_s=1e9, reads_per_ch=rpc)
    res = {"workload": workload, "team": os.environ.get("UNC_RT_TEAM", "8 (default)"), "channels": 512, "reads_per_channel": rpc, "rounds": rounds,
           "reads_finished_on_the_gpu": out["config"]["reads_finished"], "round_ms": out["config"]["latency_ms"],
           "verify": out.get("verify"), "reference": (out.get("cpu_baseline") or {}).get("kind"), "wall_s": round(time.time() - t0, 1)}
    print(json.dumps(res))
    sys.exit(1 if (out.get("verify") or {}).get("paf_mismatches", 1) else 0)
n_chk = int(sys.argv[2]) if len(sys.argv) > 2 else 10240
threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(64, len(os.sched_getaffinity(0)))
n = {"ecoli": 50000, "chr20": 200000, "grch38": 250000}[workload]
n = int(os.environ.get("SWEEP_BATCH", n))
assert pyref.available(), "oracle/_ref/libunc_ref.so did not travel"

t0 = time.time()
pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
ix = capi.Index(pre)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
m = capi.Mapper(ix)
hits = m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
k_map_ms = m.last_timing()[1]
names = ix.seq_names()
print(f"[sweep {time.time() - t0:5.0f} s] {workload}: {n} reads mapped on the GPU, k_map {k_map_ms:.0f} ms, {100 * hits['mapped'].mean():.1f} % mapped", file=sys.stderr, flush=True)

n_chk = min(n_chk, n)
pick = np.sort(np.random.default_rng(20260927).choice(n, size=n_chk, replace=False))
off = sim["offsets"].astype(np.int64)
ln = off[1:] - off[:-1]
off_s = np.concatenate(([0], np.cumsum(ln[pick]))).astype(np.uint64)
raw = np.empty(int(off_s[-1]), dtype=np.int16)
for j, i in enumerate(pick):
    raw[int(off_s[j]):int(off_s[j + 1])] = sim["signal"][int(off[i]):int(off[i + 1])].cpu().numpy()
m.close(); ix.close(); del sim
torch.cuda.empty_cache()
sig = po.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
del raw
pyref.init(pre)
pyref.set_sort_mode(pyref.SORT_STABLE)
print(f"[sweep {time.time() - t0:5.0f} s] reference index loaded; {n_chk} reads on {threads} threads ...", file=sys.stderr, flush=True)
ref, secs = pyref.map_batch(sig, off_s, threads)
print(f"[sweep {time.time() - t0:5.0f} s] reference done: {n_chk / secs:.1f} reads/s", file=sys.stderr, flush=True)

FIELDS = ("mapped", "fwd", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "rf_len", "matches", "n_events", "event_i", "n_nbr", "n_sa", "n_lf")
bad = {}
n_mapped = 0
for j, i in enumerate(pick):
    g, r = hits[i], ref[j]
    n_mapped += int(r.mapped)
    diffs = []
    for f in FIELDS:
        if f in ("fwd", "rd_st", "rd_en", "rf_st", "rf_en", "rf_len", "matches") and not r.mapped and not g["mapped"]:
            continue        # (undefined for an unmapped read on both sides)
        if int(g[f]) != int(getattr(r, f)):
            diffs.append(f)
    if np.float32(g["mean_event_len"]).tobytes() != np.float32(r.mean_event_len).tobytes():
        diffs.append("mean_event_len")
    if r.mapped and g["mapped"] and names[int(g["rid"])] != r.rf_name.decode():
        diffs.append("rf_name")
    if g["status"]:
        diffs.append("status")
    if diffs:
        bad[int(i)] = diffs
# A mismatch can be the REFERENCE's doing: map_batch gives every host thread ONE Mapper that maps read after read, and sources_added_
# -- the one piece of Mapper state new_read() does not reset (mapper.cpp:88,612-623) -- leaks from a read that filled max_paths into
# the thread's next read, whichever that happens to be.  The device maps every read as a fresh Mapper does (the batch path's documented
# order).  So every mismatching read is mapped once more by a FRESH reference Mapper: what agrees then was the
# carry-over (counted, listed), what still differs is a defect.
carried, defects = {}, {}
for i, diffs in bad.items():
    j = int(np.searchsorted(pick, i))
    r2 = pyref.Mapper().map_read(sig[int(off_s[j]):int(off_s[j + 1])])
    g = hits[i]
    still = [f for f in FIELDS if not (f in ("fwd", "rd_st", "rd_en", "rf_st", "rf_en", "rf_len", "matches") and not r2.mapped and not g["mapped"])
             and int(g[f]) != int(getattr(r2, f))]
    if np.float32(g["mean_event_len"]).tobytes() != np.float32(r2.mean_event_len).tobytes():
        still.append("mean_event_len")
    (defects if still or g["status"] else carried)[i] = still or diffs
out = {"workload": workload, "batch_reads": n, "reads_checked": n_chk,
       "mismatches_explained_by_the_reference_threads_carried_sources_added": len(carried), "carried_reads": dict(list(carried.items())[:16]),
       "mismatches_against_a_fresh_reference_mapper": len(defects), "defects": dict(list(defects.items())[:16]), "drawn": "at random across the batch, seed 20260927",
       "fields": list(FIELDS) + ["mean_event_len (bit pattern)", "rf_name", "status == 0"],
       "mismatching_reads": len(bad), "first_mismatches": dict(list(bad.items())[:16]),
       "mapped_by_reference": n_mapped, "reference": "oracle/_ref (the reference's sources compiled in place), stable tie order",
       "reference_threads": threads, "reference_reads_per_sec": round(n_chk / secs, 2), "reference_seconds": round(secs, 1),
       "gpu_k_map_ms_whole_batch": round(k_map_ms, 1), "wall_s": round(time.time() - t0, 1)}
print(json.dumps(out))
sys.exit(1 if defects else 0)
