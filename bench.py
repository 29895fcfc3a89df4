#!/usr/bin/env python3
"""bench.py -- reads mapped/sec of the MI355X hot path (BASELINE.json metric).

Headline = BASELINE config 2 (E. coli 4.6 Mb, 50 k synthetic r9.4.1 reads, 1 GPU).  One step = one pass of the whole hot
path (event detection -> normalisation -> match -> FM path forest -> seed clustering -> PAF coordinates) over one batch
of reads that is already resident in HBM.  `--gpus N` runs as N ranks (started by the driver's torch.distributed.run, or by this script);
reads shard across ranks (index replicated per GPU, no data-path collective), per-GPU work fixed => weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      k_map: algorithmic bytes per launch / HIP-event launch time against the 8 TB/s HBM peak
  cpu_baseline  the reference's own object code (oracle/_ref) on this box's host cores: thread-count sweep, best N,
                mean / median ms per read, the as-shipped MapPool hand-shake ("B2"), PAF mismatches against the GPU
  verify        sha256 of the hits of every timed step (all equal), reads checked against the CPU reference
  secondary     N = 1: the same measurement on `grch38_syn` (BASELINE config 4's reference, one GPU's share), `chr20_syn`
                (config 3, 200 k reads), each with roofline, sampled cpu_baseline and PAF check, and the realtime rounds
                (config 5); N > 1: config 4 itself (every rank 250 k reads on its own copy of the GRCh38 index)
`python bench.py --gpus N` without a launcher starts the N ranks itself (launch_ranks).
See DESIGN.md "Measurement" for how the numbers are derived.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
T_START = time.time()

WORKLOAD_TEXT = {
    "ecoli": "E. coli 4.6 Mb synthetic ref (ecoli_syn seed 1)",
    "chr20": "repeat-masked chr20-sized synthetic ref (chr20_syn 64.4 Mb, seed 2, 30% N-runs)",
    "hs400": "one eighth of a masked GRCh38-sized synthetic ref (hs400_syn: 8 contigs, 400 Mb, seed 3, 30% N-runs)",
    "grch38": "masked GRCh38-sized synthetic ref (grch38_syn: 24 contigs, 3.1 Gb, seed 3, 30% N-runs; seq_len 6.2 G > 2^32)",
    "example": "the reference's bundled example index (10 kb; plumbing only)",
}
READS_TEXT = ", synthetic r9.4.1 reads (3600 bases ~ 32k samples, 10% off-target), all reference defaults"


def log(*a):
    print("[bench %6.0f s]" % (time.time() - T_START), *a, file=sys.stderr, flush=True)


def ensure_index(cache, rank, barrier, workload, device, lib=None):
    """SURVEY 8(d) references -> BWA-format index + `.uncl` of THIS reference (`uncalled index`), built by rank 0."""
    from uncalled_amd.build_index import build_from_codes, masked_synthetic_genome, synthetic_genome
    if workload == "example":
        from uncalled_amd.build_index import encode_contigs, read_fasta
        prefix = ROOT / "tests" / "golden" / "example_index" / "example_ref"
        names, _, seqs = read_fasta(str(prefix) + ".fa")
        codes = encode_contigs(seqs)[0]
        return prefix, codes, [len(x) for x in seqs]
    if workload == "chr20":
        prefix = cache / "chr20_syn"
        names, lens, codes, holes, n_ambs = masked_synthetic_genome(1, 64444167, seed=2, name="chr20_syn")
    elif workload == "hs400":
        prefix = cache / "hs400_syn"
        names, lens, codes, holes, n_ambs = masked_synthetic_genome(8, 400000000, seed=3, name="hs400_syn")
    elif workload == "grch38":
        # seq_len 6.2 G, past the 2^31 limit of the other builders: uncalled_amd/build_index_big.py (chunked suffix sort)
        from uncalled_amd.build_index_big import big_masked_genome
        prefix = cache / "grch38_syn"
        names, lens, codes, holes, n_ambs = big_masked_genome(24, 3100000000, seed=3, name="grch38_syn")
    else:
        prefix = cache / "ecoli_syn"
        names, lens, codes = synthetic_genome(1, 4641652, seed=1)
        holes, n_ambs = (), None
    if rank == 0 and not (Path(str(prefix) + ".sa").exists() and Path(str(prefix) + ".uncl.ok").exists()):
        cache.mkdir(parents=True, exist_ok=True)
        t0 = time.time()
        if workload == "grch38":
            from uncalled_amd.build_index_big import build_from_codes_big
            build_from_codes_big(prefix, names, [""] * len(names), lens, codes, holes, n_ambs, device=device, verbose=True)
        else:
            build_from_codes(prefix, names, [""] * len(names), lens, codes, holes, n_ambs,
                             sa_device=device if workload in ("chr20", "hs400") else None)
        log(f"{workload}: index built in {time.time() - t0:.0f} s")
        # `uncalled index`: thresholds for THIS reference (self-alignment on the GPU + IndexParameterizer, preset
        # "default" = tgt_speed 115, scripts/uncalled:58); build_from_codes left the example's vector as a placeholder
        from uncalled_amd import capi
        from uncalled_amd.index_params import parameterize
        t0 = time.time()
        tmp_ix = capi.Index(prefix, device=int(str(device).split(":")[-1]) if "cuda" in str(device) else 0, lib=lib)
        parameterize(tmp_ix, prefix)
        tmp_ix.close()
        Path(str(prefix) + ".uncl.ok").write_text("parameterised\n")
        log(f"{workload}: index loaded + parameterised in {time.time() - t0:.0f} s")
    barrier()
    return prefix, codes, lens


def algorithmic_bytes(hits, offsets):
    """per read 2*S + 128*N_nbr + 64*N_lf + 8*N_sa + 64; split per kernel in DESIGN.md."""
    S = (offsets[1:] - offsets[:-1]).astype(np.float64)
    ev_bytes = 2.0 * S + 4.0 * hits["n_events"] + 24.0
    map_bytes = 4.0 * hits["event_i"] + 128.0 * hits["n_nbr"] + 64.0 * hits["n_lf"] + 8.0 * hits["n_sa"] + 64.0
    return float(ev_bytes.sum()), float(map_bytes.sum())


def cpu_baseline(prefix, raw_host, offsets, calib, hits_gpu, budget_s=80.0, sweep=(1, 16, 64, 128, 256)):
    """The CPU path on this box's host cores over a BOUNDED sample of the same reads (rank 0, N=1 only), and the GPU's PAF
    columns checked against it.  oracle/_ref (the reference's own object code) when it travelled, else the C restatement.
    The stated baseline is the BEST thread count of a sweep, run for >= 30 s (budget permitting)."""
    from oracle import pyoracle as po
    from oracle import pyref
    from uncalled_amd import capi
    aff = len(os.sched_getaffinity(0))
    kind = "reference" if pyref.available() else "port"
    n_avail = offsets.size - 1
    sig = po.calibrate(raw_host[:int(offsets[n_avail])], float(calib["range"][0]), float(calib["offset"][0]), float(calib["digitisation"][0]))
    oix = po.Index(prefix)
    names = oix.ref_names()
    if kind == "reference":
        pyref.init(prefix)

    def run(n, threads, pool=False):
        n = int(max(1, min(n, n_avail)))
        off = offsets[:n + 1]
        if kind == "reference":
            hits, secs = pyref.map_batch(sig, off, threads, pool=pool)
            cols = [h.paf_cols() for h in hits]
            ms = np.array([h.map_ms for h in hits])
        else:
            hits, secs = po.map_batch(oix, sig, off, threads)
            cols = [po.hit_paf_cols(h, names) for h in hits]
            ms = None
        return n, secs, cols, ms

    t_begin = time.time()
    # one thread first: the per-core rate everything else is sized from
    n1, s1, cols1, ms1 = run(8, 1)
    rate1 = n1 / s1
    checked = {i: c for i, c in enumerate(cols1)}
    per_leg = max(3.0, budget_s / 16.0)
    sweep_out = {"1": round(rate1, 2)}
    best_n, best_rate = 1, rate1
    for t in [t for t in sweep if 1 < t <= aff] + ([aff] if aff not in sweep and aff > 1 else []):
        # sized as if the extra threads bought nothing (they rarely buy much: every Mapper drags 2.8 MB of path buffers
        # through the caches), at least four reads per thread so that the dynamic queue can balance
        n = max(4 * t, int(best_rate * per_leg))
        n, secs, cols, _ = run(n, t)
        rate = n / secs
        sweep_out[str(t)] = round(rate, 2)
        checked.update({i: c for i, c in enumerate(cols)})
        if rate > best_rate:
            best_n, best_rate = t, rate
    # the stated figure: best N for >= 30 s (or what is left of the budget, never below 10 s)
    left = max(10.0, min(35.0, budget_s - (time.time() - t_begin) - 10.0))
    n, secs, cols, ms = run(int(best_rate * left), best_n)
    checked.update({i: c for i, c in enumerate(cols)})
    out = dict(value=n / secs, unit="reads/s", cores=best_n, kind=kind,
               sample=f"{n} reads on {best_n} threads ({secs:.1f} s), tight new_read->map_read loop "
                      f"(SURVEY 8d B1); best of the thread sweep",
               seconds=secs, host_threads_available=aff, os_cpu_count=os.cpu_count(),
               thread_sweep_reads_per_sec=sweep_out, one_thread_reads_per_sec=rate1)
    if ms is not None:
        out["ms_per_read"] = {"mean": float(ms.mean()), "median": float(np.median(ms)), "p95": float(np.percentile(ms, 95)),
                              "note": "Mapper::new_read + map_read of one read on its thread (the PAF `mt` tag)"}
        # B2: the as-shipped MapPool hand-shake (10 ms polling sleeps), same thread count, smaller sample
        nb, sb, colsb, _ = run(int(best_rate * 8.0), best_n, pool=True)
        out["b2_mappool_reads_per_sec"] = nb / sb
        out["b2_note"] = f"{nb} reads through the MapPool hand-shake of map_pool.cpp:45-69,130-158 (10 ms sleeps), {best_n} threads"
    mism = [i for i, c in checked.items() if capi.hit_paf_cols(hits_gpu[i], names) != c]
    out["paf_reads_checked"] = len(checked)
    out["paf_mismatches_vs_gpu"] = len(mism)
    if mism:
        out["paf_mismatch_reads"] = mism[:16]
    out["tie_order_note"] = ("children tying on (fm_range, seed_prob) are ordered by creation in oracle and kernels alike; "
                             "upstream's unstable pdqsort (mapper.cpp:531) is not available here, oracle/shim uses std::stable_sort")
    if kind == "reference":
        # how much rides on that convention: the same reads under the two other tie orders of oracle/shim/pdqsort.h -- pattern-defeating
        # quicksort restated from its published algorithm, and creation order REVERSED -- with the ties counted by the shim
        n_t = int(max(8, min(n_avail, best_rate * 5.0)))
        legs = {}
        for name, mode in (("stable", pyref.SORT_STABLE), ("pdqsort_restated", pyref.SORT_PDQ_RESTATED), ("reversed_ties", pyref.SORT_REVERSED_TIES)):
            pyref.set_sort_mode(mode)
            pyref.sort_stats(reset=True)
            _, _, cols_m, _ = run(n_t, best_n)
            legs[name] = (cols_m, pyref.sort_stats())
        pyref.set_sort_mode(pyref.SORT_STABLE)
        base_cols, (n_sorts, n_tie_ev, n_tie_pairs) = legs["stable"]
        out["tie_order"] = {
            "reads": n_t, "events_with_children": n_sorts, "events_with_a_tie": n_tie_ev, "tied_adjacent_pairs": n_tie_pairs,
            "paf_lines_differing_from_stable": {k: sum(1 for a, b in zip(base_cols, v[0]) if a != b) for k, v in legs.items() if k != "stable"},
            "ties_seen": {k: {"events_with_a_tie": v[1][1], "tied_adjacent_pairs": v[1][2]} for k, v in legs.items()},
            "note": "oracle/_ref with the tie order of mapper.cpp:531 switched at run time (ref_set_sort_mode); the restated pdqsort is "
                    "not pinned against upstream's object code"}
    out["sources_added_note"] = ("sources_added_ starts clear for every read on the batch path; the reference leaks it from one read to the "
                                 "next on the same thread (mapper.cpp:88,547,612-623), which only matters after a read that filled max_paths and "
                                 "is order-dependent with -t > 1 (the chunked path, where a channel is one Mapper, reproduces the carry-over)")
    return out


def cpu_baseline_realtime(prefix, host_sig, sig_off, sel, got, chunk_len):
    """Config 5's CPU leg: the reads some channels finished during the GPU run, fed to the CPU chunk path in the same
    per-channel order (the reference's own Mapper::new_read(Chunk&) / add_chunk / process_chunk / map_chunk when oracle/_ref
    travelled, else the C restatement), one Mapper per channel on host threads; PAF columns compared with the GPU's."""
    from oracle import pyoracle as po
    from oracle import pyref
    import threading
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    kind = "reference" if pyref.available() else "port"
    if kind == "reference":
        pyref.init(prefix)
    oix = po.Index(prefix)
    names_cpu = oix.ref_names()
    by_ch = {}
    for j, (c, r_i) in enumerate(sel.tolist()):
        by_ch.setdefault(c, []).append((r_i, j))
    chans = sorted(by_ch)
    state = {"checked": 0, "chunks": 0, "mism": []}
    lock = threading.Lock()

    def work(cs):
        for c in cs:
            mp = pyref.Mapper() if kind == "reference" else po.Mapper(oix)
            for r_i, j in by_ch[c]:
                sig = po.calibrate(host_sig[int(sig_off[j]):int(sig_off[j + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
                hit, used = (mp.chunk_read(sig, chunk_len, r_i) if kind == "reference" else mp.chunk_read(sig, chunk_len))
                cols = list(hit.paf_cols() if kind == "reference" else po.hit_paf_cols(hit, names_cpu))
                with lock:
                    state["checked"] += 1
                    state["chunks"] += int(used)
                    if [str(x) for x in got[j]] != [str(x) for x in cols]:
                        state["mism"].append((c, r_i))
    n_thr = max(1, min(len(chans), len(os.sched_getaffinity(0)), 64))
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(chans[i::n_thr],)) for i in range(n_thr)]
    [t.start() for t in th]
    [t.join() for t in th]
    secs = time.perf_counter() - t0
    return {"verify": {"reads_checked": state["checked"], "paf_mismatches": len(state["mism"]), "channels": len(chans), "kind": kind,
                       "mismatch_reads": state["mism"][:8]},
            "cpu_baseline": {"value": state["chunks"] / secs if secs > 0 else None, "unit": "chunks/s", "cores": n_thr, "kind": kind,
                             "sample": f"the reads {len(chans)} channels finished in the run ({state['chunks']} chunks, {secs:.1f} s), one Mapper per "
                                       f"channel on {n_thr} host threads: the per-chunk work of RealtimePool's mapper threads "
                                       "(realtime_pool.cpp:145-261) without its polling; 512 channels hand over 512 chunks per second of signal"}}


def cpu_baseline_realtime_subprocess(prefix, host_sig, sig_off, sel, got, chunk_len):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(prefix="unc_cpu_leg_") as d:
        f = Path(d) / "leg_rt.npz"
        np.savez(f, sig=host_sig, sig_off=sig_off, sel=sel, got=json.dumps(got), prefix=str(prefix), chunk_len=chunk_len)
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-leg-rt", str(f)], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("realtime CPU leg failed: " + r.stderr[-600:])
        return json.loads(lines[-1])


def cpu_leg_rt_main(path):
    d = np.load(path, allow_pickle=False)
    print(json.dumps(cpu_baseline_realtime(str(d["prefix"]), d["sig"], d["sig_off"], d["sel"], json.loads(str(d["got"])), int(d["chunk_len"]))))


def cpu_baseline_subprocess(prefix, raw_host, offsets, calib, hits_gpu, budget_s):
    """cpu_baseline in a process of its own: the reference keeps its index in process-global statics (mapper.hpp:80-85, one
    index per process), so every reference index of a bench run needs a fresh process."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(prefix="unc_cpu_leg_") as d:
        f = Path(d) / "leg.npz"
        np.savez(f, raw=raw_host, offsets=offsets, calib=calib, hits=hits_gpu, prefix=str(prefix), budget=budget_s)
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-leg", str(f)], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("CPU baseline process failed: " + r.stderr[-600:])
        return json.loads(lines[-1])


def cpu_leg_main(path):
    d = np.load(path, allow_pickle=False)
    out = cpu_baseline(str(d["prefix"]), d["raw"], d["offsets"], d["calib"], d["hits"], budget_s=float(d["budget"]))
    print(json.dumps(out))


def a_reads(a, workload):
    return {"ecoli": a.reads, "chr20": a.chr20_reads, "grch38": a.grch38_reads}.get(workload, a.reads)


KERNEL_SOURCES = ("k_map.hip", "map_sort.h", "map_sort_wide.h", "map_tracker.h", "map_lds.h", "map_sched.h", "fm_dev.h", "wave_prims.h",
                  "unc_dev_types.h")


def kernel_source_hash(root=None):
    """sha256 over the sources k_map is compiled from (in a fixed order): what a committed counter pass is keyed by.  The library
    travels with the tree it was built from, so the sources name the kernel that runs; tools/dev/summarise_pmc.py and
    summarise_sq.py write the same hash into the records they produce."""
    import hashlib
    h = hashlib.sha256()
    base = Path(root) if root else ROOT / "uncalled_amd" / "csrc"
    for name in KERNEL_SOURCES:
        h.update(name.encode() + b"\0" + (base / name).read_bytes() + b"\0")
    return h.hexdigest()


def pmc_record(kind, workload, a, profiles=None):
    """The newest committed counter record (kind: 'k_map' = FETCH / WRITE_SIZE, 'sq_summary' = SQ counters) of this workload that was
    taken on THIS kernel and THIS batch size, else (None, why not): a record of another kernel is not a measurement of this one."""
    profiles = Path(profiles) if profiles else ROOT / "profiles"
    cands = sorted(profiles.glob(f"r[0-9][0-9]_pmc_{kind}_{workload}.json"), reverse=True)
    if not cands:
        return None, "no PMC pass committed for this workload"
    want = kernel_source_hash()
    why = []
    for f in cands:
        d = json.loads(f.read_text())
        if d.get("workload", "ecoli") != workload or int(d.get("reads_per_launch", -1)) != a_reads(a, workload):
            why.append(f"{f.name}: collected on {d.get('workload')} / {d.get('reads_per_launch')} reads")
            continue
        if d.get("kernel_source_sha256") != want:
            why.append(f"{f.name}: taken on another kernel (source hash {str(d.get('kernel_source_sha256'))[:12]}, running {want[:12]})")
            continue
        d["_file"] = f.name
        return d, None
    return None, "; ".join(why)[:300]


def measured_traffic(a, workload):
    """HBM bytes per launch of k_map from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot be collected
    from inside this process): used only when they were taken on this workload, this batch size and THIS kernel (the record carries the
    hash of the kernel's sources), else null with the reason."""
    d, why = pmc_record("k_map", workload, a)
    if d is None:
        return None, why
    pmc = Path(d["_file"])
    src = f"profiles/{pmc.name} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes; calibrated, see the file)"
    if d.get("kernel_state"):
        src += "; " + d["kernel_state"]
    return float(d["hbm_bytes_per_launch"]), src


def issue_roofline(a, workload, launch_ms, clock_hz):
    """Issue side of k_map: wave-instructions of one launch (SQ_INSTS of the committed rocprofv3 pass on this workload, batch
    size and kernel -- the count is a property of kernel + batch, the duration is this run's) / (1024 SIMDs x clock x launch time)."""
    d, _ = pmc_record("sq_summary", workload, a)
    if d is None:
        return None
    sq = Path(d["_file"])
    insts = d.get("counters", {}).get("SQ_INSTS")
    if not insts:
        return None
    c = d["counters"]
    util = insts / (1024 * clock_hz * launch_ms * 1e-3)
    out = {"wave_instructions_per_launch": insts, "simds": 1024, "clock_ghz": clock_hz * 1e-9, "utilisation": util,
           "source": f"profiles/{sq.name} (rocprofv3 --pmc SQ_INSTS on this batch) / this run's launch time",
           # (SQ_ACTIVE_INST_VMEM reads 0 on this build: the dead counter of rounds 3-4's summaries is not passed on)
           "wave_cycle_shares": {k: round(v, 4) for k, v in d.get("derived", {}).items()
                                 if k.startswith("wave_cycle_share_") and k != "wave_cycle_share_active_inst_vmem"},
           "per_read": {k[:-9]: round(v) for k, v in d.get("derived", {}).items() if k.endswith("_per_read")}}
    if c.get("SQ_INSTS_VALU"):
        out["valu_utilisation"] = c["SQ_INSTS_VALU"] / (1024 * clock_hz * launch_ms * 1e-3)
    if c.get("SQ_INSTS_VALU"):
        # how busy the vector ALUs are.  Round 5 took 4 cycles per wave64 instruction (SQ_ACTIVE_INST_VALU counts quad-cycles, one per
        # instruction) and read 55 %.  Measured in round 6 (tools/dev/ubench_valu.hip, profiles/r06_ubench_valu.log): four wavefronts
        # of independent v_add_u32 / v_fma_f32 on one SIMD retire one instruction per 2.1 - 2.3 cycles (MI355X_MICROARCH.md: SIMD-32,
        # two cycles per wave64 instruction) -- the quad-cycle counter rounds every instruction up to four.  Two cycles each:
        out["valu_pipe_busy"] = 2.0 * c["SQ_INSTS_VALU"] / (1024 * clock_hz * launch_ms * 1e-3)
        out["valu_pipe_busy_note"] = "2 cycles x SQ_INSTS_VALU / (1024 SIMDs x clock x launch): profiles/r06_ubench_valu.log"
    return out


def run_workload(a, workload, n_reads, steps, warmup, rank, world, local_rank, dist, barrier, cache, lib, dev_name, extras, cpu_budget,
                 warmup_reads=None):
    """index (built once, cached) -> reads synthesised in HBM -> warm-up + timed steps -> result dict"""
    import torch
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from tools.simulate_reads_torch import simulate_reads_torch
    from uncalled_amd import capi
    have_gpu = dev_name != "cpu"
    prefix, codes, lens = ensure_index(cache, rank, barrier, workload, dev_name, lib)
    ix = capi.Index(prefix, device=local_rank if have_gpu else 0, lib=lib)
    if have_gpu:
        torch.cuda.empty_cache()                 # blocks torch still caches from the index build: the mapper sizes its pool by what is free
    kw = dict(pool_chunks=a.pool_chunks, n_slots=getattr(a, "slots", 0))
    if not have_gpu:
        kw.update(n_slots=4, n_waves=2)          # lanesim plumbing test
    mapper = capi.Mapper(ix, **kw)
    # this rank's shard of the read set: reads are independent units, sharded by rank with distinct seeds
    sim = simulate_reads_torch(codes, lens, n_reads, seed=42 + rank, device=dev_name)
    del codes
    offsets = sim["offsets"]
    calib = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    raw_ptr = sim["signal"].data_ptr()
    stream = torch.cuda.current_stream().cuda_stream if have_gpu else None

    def sync():
        if have_gpu:
            torch.cuda.synchronize()

    def one_step(nr=None):
        nr = n_reads if nr is None else min(nr, n_reads)
        if have_gpu:
            return mapper.map_batch_device(raw_ptr, offsets[:nr + 1], calib[:nr], stream=stream)
        return mapper.map_batch(sim["signal"].numpy()[:int(offsets[nr])], offsets[:nr + 1], calib[:nr])

    # Two mappers over the one index, batch k + 1 begun on the second while batch k's last long reads finish on the first
    # (unc_map_batch_begin / _end, include/uncalled_hip.h): a persistent launch ends with a few wavefronts on a few long reads
    # (wavefronts alive 94 % of a 50 k-read E. coli launch), and the next batch moves into the compute units as they fall idle.  Every
    # step is still one whole pass of the path over one batch, each batch's hits are complete when its _end returns, and all K are
    # collected inside the timed region.  Where a second mapper's scratch does not fit beside the first (GRCh38) the steps run one
    # after the other as before.
    mapper2 = None
    if have_gpu and a.pipeline and steps > 1 and hasattr(mapper.L, "unc_map_batch_begin"):
        free_b, _tot = torch.cuda.mem_get_info(local_rank)
        if free_b > 1.25 * mapper.device_bytes():
            mapper2 = capi.Mapper(ix, **kw)
    pair = [mapper, mapper2] if mapper2 is not None else [mapper]

    # (the first mapper's stream outranks the second's: two batches begun at the same moment -- the first two of the timed region --
    # would otherwise share the compute units from the start and end together, tail beside tail; later batches are begun while
    # the one before holds every wavefront slot, and wait for slots whatever their rank)
    pstreams = [torch.cuda.Stream(device=local_rank, priority=-1), torch.cuda.Stream(device=local_rank, priority=0)] if mapper2 is not None else []

    def begin(mp, nr=None):
        nr = n_reads if nr is None else min(nr, n_reads)
        mp.begin_batch(raw_ptr, offsets[:nr + 1], calib[:nr], on_device=True, stream=pstreams[pair.index(mp)].cuda_stream)

    hits = None
    for _ in range(warmup):
        for mp in pair:
            hits = mp.map_batch_device(raw_ptr, offsets[:min(warmup_reads or n_reads, n_reads) + 1], calib[:min(warmup_reads or n_reads, n_reads)]) \
                if have_gpu else one_step(warmup_reads)      # (secondary blocks warm up on a prefix of their batch: code, TLBs, first touch)
    sync()
    barrier()
    t0 = time.perf_counter()
    ms_ev, ms_map, kept, step_ms, windows = [], [], [], [], []
    if mapper2 is None:
        for _ in range(steps):
            ts = time.perf_counter()
            hits = one_step()                  # (returns the hits: the step is complete on the device when it returns)
            step_ms.append(1e3 * (time.perf_counter() - ts))
            e, m = mapper.last_timing()
            ms_ev.append(e)
            ms_map.append(m)
            kept.append(hits)
    else:
        ts = time.perf_counter()
        begin(pair[0])
        for k in range(steps):
            if k + 1 < steps:
                begin(pair[(k + 1) % 2])
            hits = pair[k % 2].end_batch()     # (batch k complete: its hits are on the host)
            tn = time.perf_counter()
            step_ms.append(1e3 * (tn - ts))
            ts = tn
            e, m = pair[k % 2].last_timing()
            ms_ev.append(e)
            ms_map.append(m)
            if hasattr(mapper.L, "unc_mapper_last_window"):
                windows.append(pair[k % 2].last_window())
            kept.append(hits)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    # self-verification, outside the timed region: every step must have produced the same bytes
    digests = [capi.hits_digest(h) for h in kept]      # (every result field; map_ms is a wall-clock measurement)
    del kept
    per_rank = None
    if dist is not None:
        # MAX over ranks is the job's time; every rank's own time goes along (min / median / max): a scaling curve must be able to tell
        # one slow rank from a slow design
        t = torch.tensor([dt], dtype=torch.float64, device=dev_name if have_gpu else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        ts_ = sorted(float(x.item()) for x in allt)
        per_rank = {"min": 1e3 * ts_[0] / steps, "median": 1e3 * ts_[len(ts_) // 2] / steps, "max": 1e3 * ts_[-1] / steps,
                    "unit": "ms per step, each rank's own timed region"}
        dt = ts_[-1]
    assert len(set(digests)) == 1, f"{workload}: hits differ between timed steps: {digests}"

    wave_busy = mapper.last_wave_busy()
    remap_n, remap_ms = mapper.last_remap()
    res = {"value": n_reads * world * steps / dt, "ms_per_step": 1e3 * dt / steps, "dt": dt}
    if per_rank:
        res["per_rank"] = per_rank
    pcie, phase_share, t1_info = None, None, None
    if extras and rank == 0 and world == 1:
        if have_gpu and workload == "ecoli":
            # the boundary also takes host buffers (unc_map_batch on_device = 0): one extra, untimed step from pageable host
            # memory gives the PCIe-inclusive rate (never `value`)
            host_raw = sim["signal"].cpu().numpy()
            t1 = time.perf_counter()
            h2 = mapper.map_batch(host_raw, offsets, calib)
            pcie = n_reads / (time.perf_counter() - t1)
            assert capi.hits_digest(h2) == digests[0], "host-buffer path differs from the device-buffer path"
            del host_raw
        # `uncalled map -t 1` order: one extra, untimed pass with sources_added_ carried from read to read (UNC_ORDER_T1): what it costs
        # and how many reads it changes against the independent order of the timed steps
        if hasattr(mapper.L, "unc_mapper_set_read_order"):
            mapper.set_read_order(capi.ORDER_T1)
            t1 = time.perf_counter()
            ht = one_step()
            t1 = time.perf_counter() - t1
            n_again, rounds, ms_again = mapper.last_carry_over()
            mapper.set_read_order(capi.ORDER_INDEPENDENT)
            changed = [f for f in capi.RESULT_FIELDS if not np.array_equal(ht[f], hits[f])]
            t1_info = {"reads_mapped_again": n_again, "rounds": rounds, "ms_spent_on_them": round(ms_again, 1), "step_s": round(t1, 3),
                       "reads_whose_result_changed": int(np.any([ht[f] != hits[f] for f in capi.RESULT_FIELDS if f != "notes"], axis=0).sum()),
                       "reads_whose_paf_columns_changed": int(np.any([ht[f] != hits[f] for f in ("mapped", "fwd", "rid", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "matches")], axis=0).sum()),
                       "fields_that_changed": changed,
                       "note": "UNC_ORDER_T1 = the reads of the batch as ONE Mapper maps them back to back (parity: tests/parity_cases.py:case_read_order_t1, "
                               "oracle pinned on the reference's Mapper in tests/test_oracle.py); the timed steps use the independent order"}
            del ht
        # phase shares: one extra, untimed pass with the cycle-counting instantiation of k_map
        prof_reads = min(n_reads, 50000)
        mapper.set_profile(True)
        hp = one_step(prof_reads)
        mapper.set_profile(False)
        assert capi.hits_digest(hp) == capi.hits_digest(hits[:prof_reads]), "profiling instantiation differs from the plain one"
        pc = mapper.last_phase_cycles()
        tot_c = float(sum(pc.values())) or 1.0
        phase_share = {k: round(v / tot_c, 4) for k, v in pc.items()}
    if rank == 0:
        ev_bytes, map_bytes = algorithmic_bytes(hits, offsets)
        map_ms_each = float(np.mean(ms_map))
        map_ms = map_ms_each
        overlap = None
        if len(windows) == steps and steps > 1:
            # batches of two mappers in flight at once: the launches overlap (a launch waits for wavefront slots while its predecessor's
            # last long reads finish, then takes them over), so the time k_map holds the device PER BATCH is the union of the launches'
            # windows (HIP events on one time axis, unc_mapper_last_window) over the number of batches -- the denominator of a
            # throughput roofline; the mean length of a window (launch_ms_each: what rocprofv3 lists per dispatch) counts the shared
            # stretches twice
            iv = sorted(windows)
            union, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > cur_e:
                    union += cur_e - cur_s
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            union += cur_e - cur_s
            map_ms = union / steps
            overlap = {"launch_ms_each": map_ms_each, "union_ms": union, "launches": steps, "mean_launches_resident": sum(e_ - s_ for s_, e_ in iv) / union}
        ev_ms = float(np.mean(ms_ev))
        achieved = map_bytes / (map_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(a, workload)
        clock_hz = 2.4e9
        if have_gpu:
            clock_hz = float(getattr(torch.cuda.get_device_properties(local_rank), "clock_rate", 2.4e6)) * 1e3
        issue = issue_roofline(a, workload, map_ms, clock_hz)
        # what binds k_map: neither roofline -- its wavefronts wait on dependent scattered loads (wave_cycle_shares below)
        bound_note = ("hbm is the stated roofline; measured: issue slots %.0f %% used, waves waiting on memory %.0f %% of their cycles -> "
                      "latency of dependent scattered accesses, not bandwidth and not issue" %
                      (100 * issue["utilisation"], 100 * issue["wave_cycle_shares"].get("wave_cycle_share_wait_any", 0))) if issue else None
        res.update({
            "config": {"workload": WORKLOAD_TEXT[workload] + READS_TEXT,
                       "reads_per_gpu_per_step": n_reads, "parallelism": f"reads sharded over {world} GPU(s), index replicated",
                       "mean_ms_per_read_amortised": 1e3 * dt / (n_reads * steps),
                       "mapped_fraction": float(hits["mapped"].mean()),
                       "mean_events_per_read": float(hits["event_i"].mean()),
                       "kernel_ms": {"k_events": ev_ms, "k_map": map_ms},
                       "batches_in_flight": len(pair),
                       "step_ms": {"median": float(np.median(step_ms)), "min": float(np.min(step_ms)), "max": float(np.max(step_ms)),
                                   "k_map_min": float(np.min(ms_map)), "k_map_max": float(np.max(ms_map))},
                       "k_map_phase_cycle_share": phase_share,
                       "k_map_phase_cycle_share_source": "extra untimed pass over the first min(n, 50 000) reads, profiling instantiation of k_map",
                       "k_map_wave_busy": round(wave_busy, 4),
                       "pcie_inclusive_reads_per_sec": pcie,
                       "gpu_ms_per_read": {"mean": float(hits["map_ms"].mean()), "median": float(np.median(hits["map_ms"])),
                                           "p95": float(np.percentile(hits["map_ms"], 95)), "max": float(hits["map_ms"].max()),
                                           "note": "residence of a read on the device: first event taken up -> result written (device wall clock, "
                                                   "the PAF `mt` tag; reads share wavefronts in time slices of 1024 events, so this is not service time)"},
                       "k_map_code_object": mapper.kernel_info(),
                       "remapped_reads": {"n": remap_n, "ms": round(remap_ms, 1),
                                          "note": "reads that found the seed-cluster node pool dry, mapped again after the batch (inside the step)"},
                       "reads_in_flight": mapper.geometry(),
                       "node_pool": mapper.pool_usage(),
                       "index_seq_len": int(ix.size), "index_device_bytes": int(ix.device_bytes())},
            "roofline": {"bound": "hbm", "kernel": "k_map", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": (traffic / map_bytes) if traffic else None,
                         "issue": issue, "bound_note": bound_note,
                         "algorithmic_bytes_per_launch": map_bytes, "launch_ms": map_ms, "launch_ms_each": map_ms_each, "overlap": overlap,
                         "whole_path_bytes_per_step": ev_bytes + map_bytes,
                         "k_events": {"algorithmic_bytes_per_launch": ev_bytes, "launch_ms": ev_ms,
                                      "achieved": ev_bytes / (ev_ms * 1e-3) / 1e9 if ev_ms > 0 else None}},
            "verify": {"steps_hashed": len(digests), "all_steps_identical": True, "hits_sha256": digests[0],
                       # the only state the reference's Mapper carries from one read into the next (sources_added_, mapper.cpp:88,612-623)
                       # exists only after a read that filled its path buffer: counted per read by the kernel (unc_hit_t::notes)
                       "reads_that_filled_max_paths": int(((hits["notes"] & capi.NOTE_PATHS_FULL) != 0).sum()),
                       "reads_ending_with_sources_added_set": int(((hits["notes"] & capi.NOTE_FLAGS_LEFT) != 0).sum()),
                       "carry_over_note": "both 0: every read of the batch is mapped exactly as `uncalled map -t 1` maps it in any order",
                       "t1_order": t1_info},
        })
        if world == 1 and cpu_budget > 0:
            # the CPU leg and the PAF check run on reads SAMPLED ACROSS the batch (seeded), not on its first reads
            n_cpu = min(n_reads, 12288)
            pick = np.sort(np.random.default_rng(12345).choice(n_reads, size=n_cpu, replace=False))
            lens_ = (offsets[1:] - offsets[:-1]).astype(np.int64)
            off_s = np.concatenate(([0], np.cumsum(lens_[pick]))).astype(np.uint64)
            import torch as _t
            host_sig = np.empty(int(off_s[-1]), dtype=np.int16)
            for j, i in enumerate(pick):
                host_sig[int(off_s[j]):int(off_s[j + 1])] = sim["signal"][int(offsets[i]):int(offsets[i + 1])].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline_subprocess(prefix, host_sig, off_s, calib[pick], hits[pick], cpu_budget)
            res["cpu_baseline"]["sample"] += "; reads drawn at random across the batch (seed 12345)"
            res["verify"]["reads_checked_vs_cpu"] = res["cpu_baseline"]["paf_reads_checked"]
            res["verify"]["paf_mismatches"] = res["cpu_baseline"]["paf_mismatches_vs_gpu"]
    if mapper2 is not None:
        mapper2.close()
    mapper.close()
    ix.close()
    del sim
    if have_gpu:
        torch.cuda.empty_cache()
    return res


def realtime_workload(a, ix, prefix, codes, lens, local_rank, ref_label, steps, warmup, verify_channels=24, cpu_budget_s=20.0, reads_per_ch=6):
    """BASELINE config 5: 512 channels x 4000-sample chunks, deterministic MAP_ORD-style scheduling; one step = one
    chunk round (every channel hands over its next chunk, all chunks are mapped completely).  Latency is per round.
    verify: the reads `verify_channels` channels finished during the run, against the CPU chunk path (the reference's own
    Mapper::new_read(Chunk&) / add_chunk / process_chunk / map_chunk when oracle/_ref travelled, else the C restatement) fed
    the same reads in the same per-channel order; baseline: the same per-channel work on host threads."""
    import torch
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from tools.simulate_reads_torch import simulate_reads_torch
    from uncalled_amd import capi
    n_ch = a.channels
    sim = simulate_reads_torch(codes, lens, n_ch * reads_per_ch, seed=777, device=f"cuda:{local_rank}")
    off = sim["offsets"].astype(np.int64)
    rt = capi.Realtime(ix, n_channels=n_ch)
    chunk_len = 4000
    cur_read = [0] * n_ch
    cur_chunk = [0] * n_ch
    wrapped = [False] * n_ch
    raw_ptr = sim["signal"].data_ptr()
    lat, ms_ev, ms_map, n_chunks_done, finished = [], [], [], 0, 0
    done_hits = {}                                    # (channel, read of the channel) -> hit, first pass over the channel's reads only
    total_rounds = warmup + steps
    for rnd in range(total_rounds):
        ch = np.zeros(n_ch, dtype=capi.RT_CHUNK)
        k = 0
        for c in range(n_ch):
            if cur_read[c] >= reads_per_ch:
                cur_read[c] = 0     # replay the channel's reads: the channel never idles
                wrapped[c] = True
            r = c * reads_per_ch + cur_read[c]
            n = int(off[r + 1] - off[r])
            st = min(cur_chunk[c] * chunk_len, n)
            ln = min(chunk_len, n - st)
            fl = (capi.RT_FIRST if cur_chunk[c] == 0 else 0) | (capi.RT_LAST if st + ln >= n else 0)
            ch[k]["channel"], ch[k]["read_number"], ch[k]["flags"], ch[k]["n_samples"], ch[k]["offset"] = c, cur_read[c], fl, ln, off[r] + st
            ch[k]["calib"]["range"], ch[k]["calib"]["offset"], ch[k]["calib"]["digitisation"] = CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION
            k += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = rt.process_chunks(ch[:k], raw_ptr=raw_ptr)
        dt = time.perf_counter() - t0
        e, m = rt.last_timing()
        if rnd >= warmup:
            lat.append(dt * 1e3); ms_ev.append(e); ms_map.append(m); n_chunks_done += k
        for j in range(k):
            c = int(ch[j]["channel"])
            if res[j]["state"] == capi.RT_MAPPING:
                cur_chunk[c] += 1
            else:
                if not wrapped[c]:
                    done_hits[(c, cur_read[c])] = (res[j]["hit"].copy(), int(res[j]["state"]))
                cur_chunk[c] = 0
                cur_read[c] += 1
                finished += 1
    lat = np.array(lat)
    out = {"metric": "chunk_round_latency_ms", "value": float(lat.mean()), "unit": "ms", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": float(lat.mean()), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
           "dtype": "u64+f32/f64", "data": "synthetic",
           "config": {"workload": f"realtime: {n_ch} channels x {chunk_len}-sample chunks (1 s of signal each), "
                                  f"{WORKLOAD_TEXT[ref_label]}, MAP_ORD-style deterministic scheduling, raw signal resident in HBM",
                      "latency_ms": {"mean": float(lat.mean()), "p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)),
                                     "max": float(lat.max())},
                      "chunks_per_sec": n_chunks_done / (lat.sum() * 1e-3), "reads_finished": finished,
                      "kernel_ms": {"k_rt_events": float(np.mean(ms_ev)), "k_map": float(np.mean(ms_map))},
                      "sla": "a chunk is 1000 ms of signal; the round must finish well inside that"}}
    rt.close()
    # ---- verify + host baseline (outside the timed rounds; in a process of its own, as every CPU leg)
    if cpu_budget_s > 0:
        chans = sorted({c for (c, _) in done_hits})[:verify_channels]
        names_dev = ix.seq_names()
        sel, got, sig_parts, sig_off = [], [], [], [0]
        for c in chans:
            for r_i in range(reads_per_ch):
                if (c, r_i) not in done_hits:
                    break                       # the channel's later reads were not finished in the run: stop at the first gap
                r = c * reads_per_ch + r_i
                sel.append((c, r_i))
                got.append(list(capi.hit_paf_cols(done_hits[(c, r_i)][0], names_dev)))
                sig_parts.append(sim["signal"][int(off[r]):int(off[r + 1])].cpu().numpy())
                sig_off.append(sig_off[-1] + int(off[r + 1] - off[r]))
        if sel:
            leg = cpu_baseline_realtime_subprocess(prefix, np.concatenate(sig_parts), np.array(sig_off, dtype=np.uint64),
                                                   np.array(sel, dtype=np.int64), got, chunk_len)
            out["verify"] = leg["verify"]
            out["cpu_baseline"] = leg["cpu_baseline"]
    return out


def _r(x, sig=6):
    """floats of the printed line: 6 significant digits (a non-finite one becomes null: the line must stay strict JSON)"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    return x


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d and d[k] is not None}


def compact_block(blk, top=True):
    """The numbers of one measured block that go into the printed line (everything else: bench_detail.json)."""
    if "error" in blk or "skipped" in blk:
        return {k: str(v)[:160] for k, v in blk.items()}
    out = _pick(blk, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")
    cfg = blk.get("config", {})
    c = _pick(cfg, "reads_per_gpu_per_step", "mapped_fraction", "mean_events_per_read", "chunks_per_sec", "reads_finished", "batches_in_flight")
    if "kernel_ms" in cfg:
        c["kernel_ms"] = {k: _r(v, 5) for k, v in cfg["kernel_ms"].items()}
    if "latency_ms" in cfg:
        c["latency_ms"] = {k: _r(v, 4) for k, v in cfg["latency_ms"].items()}
    if "step_ms" in cfg:
        c["step_ms"] = {k: _r(v, 5) for k, v in cfg["step_ms"].items()}
    if cfg.get("remapped_reads"):
        c["remapped_reads"] = cfg["remapped_reads"]["n"]
    if cfg.get("node_pool"):
        c["node_pool"] = cfg["node_pool"]
    out["config"] = c
    if "roofline" in blk:
        keys = ("achieved", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch", "launch_ms", "launch_ms_each")
        out["roofline"] = _pick(blk["roofline"], *((("bound", "kernel", "peak", "unit") + keys) if top else keys))
        out["roofline"].setdefault("traffic", None)
        iss = blk["roofline"].get("issue") or {}
        if iss.get("valu_pipe_busy") is not None:
            out["roofline"]["valu_pipe_busy"] = _r(iss["valu_pipe_busy"], 3)       # the physically binding side: issue, not bytes
        if iss.get("wave_cycle_shares", {}).get("wave_cycle_share_wait_any") is not None:
            out["roofline"]["wave_wait_share"] = _r(iss["wave_cycle_shares"]["wave_cycle_share_wait_any"], 3)
    if blk.get("ms_per_read"):
        out["ms_per_read"] = {k: _r(v, 4) for k, v in blk["ms_per_read"].items()}
    if blk.get("per_rank"):
        out["per_rank"] = {k: _r(v, 5) for k, v in blk["per_rank"].items() if k != "unit"}
    if "cpu_baseline" in blk:
        out["cpu_baseline"] = _pick(blk["cpu_baseline"], "value", "unit", "cores", "kind", "paf_reads_checked", "paf_mismatches_vs_gpu")
        out["cpu_baseline"]["sample"] = str(blk["cpu_baseline"].get("sample", ""))[:120 if top else 48]
    if "verify" in blk:
        v = blk["verify"]
        out["verify"] = _pick(v, "steps_hashed", "all_steps_identical", "reads_checked_vs_cpu", "paf_mismatches", "reads_checked")
        if "hits_sha256" in v:
            out["verify"]["hits_sha16"] = v["hits_sha256"][:16]
    return out


LINE_LIMIT = 8192       # the driver keeps a bounded tail of stdout: the line must fit in it whole (round 4's 21.7 KB line did not parse)


def emit(out):
    """Full record -> bench_detail.json (path on stderr); ONE compact JSON line -> stdout."""
    detail = Path(os.environ.get("UNC_BENCH_DETAIL", ROOT / "gpurun_out" / "bench_detail.json"))
    try:
        detail.parent.mkdir(parents=True, exist_ok=True)
        detail.write_text(json.dumps(out, indent=1))
        log(f"full record (prose notes, thread sweeps, phase shares, code object): {detail}")
    except OSError as e:
        log(f"could not write {detail}: {e}")
    line = compact_block(out)
    line["config"]["workload"] = out["config"]["workload"][:200]
    line["config"]["parallelism"] = out["config"].get("parallelism")
    for k in ("higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        line[k] = out[k]
    if "secondary" in out:
        line["secondary"] = {k: compact_block(v, top=False) for k, v in out["secondary"].items()}
    if "bench_wall_s" in out:
        line["bench_wall_s"] = _r(out["bench_wall_s"], 4)
    def dumps(x):
        return json.dumps(x, separators=(",", ":"), allow_nan=False)
    txt = dumps(line)
    if len(txt) >= LINE_LIMIT:      # never an assert at the end of a ten-minute run: shed what the detail file holds anyway
        for blk in line.get("secondary", {}).values():
            for k in ("step_ms", "node_pool", "latency_ms"):
                blk.get("config", {}).pop(k, None)
            blk.get("cpu_baseline", {}).pop("sample", None)
        txt = dumps(line)
    if len(txt) >= LINE_LIMIT:
        line["secondary"] = {k: {kk: v.get(kk) for kk in ("value", "unit", "n_gpus", "ms_per_step") if kk in v} for k, v in line.get("secondary", {}).items()}
        txt = dumps(line)
    print(txt, flush=True)
    return txt


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU; torch.distributed.run on this
    node, rendezvous on 127.0.0.1) with the same arguments, pass their output through, return their exit code.  The reference's
    parallelism is N independent workers over one read queue (src/map_pool.cpp:31-42); here a worker is a rank with its own GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py")] + sys.argv[1:]
    log(f"--gpus {n}: launching {n} ranks: {' '.join(cmd[2:9])} ...")
    rc = subprocess.run(cmd).returncode
    if rc:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None,
                    help="reads per GPU per step of the headline workload (default: 50 000 = BASELINE config 2; UNC_BENCH_READS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool-chunks", type=int, default=0, help="chunks of the seed-cluster node pool (0 = library default)")
    ap.add_argument("--slots", type=int, default=0, help="reads in flight per mapper (0 = library default)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the extra untimed passes (PCIe-inclusive rate, phase cycle shares)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="one mapper, the steps one after the other (default: two mappers, batch k + 1 begun while batch k's tail drains)")
    ap.add_argument("--workload", choices=["ecoli", "chr20", "hs400", "grch38", "realtime", "example"], default="ecoli",
                    help="headline workload (the driver runs the default: BASELINE config 2)")
    ap.add_argument("--secondary", default=os.environ.get("UNC_BENCH_SECONDARY"),
                    help="comma list of further workloads measured after the headline ('' = none; default: realtime:ecoli, chr20, "
                         "realtime:chr20, grch38 at N = 1, grch38 -- BASELINE config 4, every rank its 250 k reads -- at N > 1)")
    ap.add_argument("--grch38-reads", type=int, default=250000, help="reads of the grch38 block (config 4: 2 M reads over 8 GPUs = 250 k per GPU)")
    ap.add_argument("--chr20-reads", type=int, default=200000, help="reads of the chr20 block (config 3)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("UNC_BENCH_BUDGET_S", 1500)),
                    help="secondary blocks are skipped once this much wall time has gone")
    ap.add_argument("--secondary-steps", type=int, default=3, help="timed steps of the chr20 / grch38 blocks (after one untimed step over the whole batch)")
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--rt-ref", choices=["ecoli", "chr20", "grch38"], default="ecoli", help="reference of the realtime workload")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)      # internal: cpu_baseline_subprocess
    ap.add_argument("--cpu-leg-rt", default=None, help=argparse.SUPPRESS)   # internal: cpu_baseline_realtime_subprocess
    a = ap.parse_args()
    if a.cpu_leg:
        return cpu_leg_main(a.cpu_leg)
    if a.cpu_leg_rt:
        return cpu_leg_rt_main(a.cpu_leg_rt)
    if a.reads is None:
        a.reads = a_reads(argparse.Namespace(reads=int(os.environ.get("UNC_BENCH_READS", 50000)), chr20_reads=a.chr20_reads,
                                             grch38_reads=a.grch38_reads), a.workload)

    if a.gpus is None:
        # started by a launcher without --gpus (`torchrun --nproc-per-node 8 bench.py`): the launcher's world is the statement
        a.gpus = int(os.environ.get("WORLD_SIZE", 1))
        if a.gpus > 1 and int(os.environ.get("RANK", 0)) == 0:
            log(f"--gpus not given: n_gpus = WORLD_SIZE = {a.gpus}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus)         # plain `python bench.py --gpus N`: one rank per GPU, this process only relays
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:      # an EXPLICIT --gpus that disagrees with the launcher: the line would state the wrong n_gpus
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (n_gpus would be wrong); drop --gpus or fix the launcher")
    # UNC_BENCH_LIB: tests point this at the lanesim build of the same sources to run the world > 1 plumbing on CPU ranks
    lib_path = os.environ.get("UNC_BENCH_LIB")
    have_gpu = lib_path is None
    from uncalled_amd import capi
    lib = capi.load(lib_path) if lib_path else None
    if have_gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        torch.cuda.set_device(local_rank)
    dev_name = f"cuda:{local_rank}" if have_gpu else "cpu"
    dist = None
    placement = None
    if world > 1:
        # each rank's host threads (this one, the library's staging / loader threads) on the NUMA node of its GPU
        from uncalled_amd.numa import pin_to_gpu_node
        placement = pin_to_gpu_node(local_rank, world) if have_gpu else None
        if placement:
            log(f"rank {rank}: host threads -> {placement}")
        import torch.distributed as dist
        backend = os.environ.get("UNC_DIST_BACKEND", "nccl" if have_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    cache = Path(os.environ.get("UNC_BENCH_CACHE", "/tmp/uncalled_amd_bench"))
    if a.workload == "realtime":
        prefix, codes, lens = ensure_index(cache, rank, barrier, a.rt_ref, dev_name)
        ix = capi.Index(prefix, device=local_rank)
        out = realtime_workload(a, ix, prefix, codes, lens, local_rank, a.rt_ref, a.steps, a.warmup,
                                cpu_budget_s=0.0 if a.no_cpu_baseline else 20.0)
        if rank == 0:
            emit(out)
        return

    extras = not a.no_profile_pass
    cpu_budget = 0.0 if (a.no_cpu_baseline or not have_gpu) else 100.0
    head = run_workload(a, a.workload, a.reads, a.steps, a.warmup, rank, world, local_rank, dist, barrier, cache, lib, dev_name,
                        extras, cpu_budget)
    out = None
    if rank == 0:
        out = {
            "metric": "reads_mapped_per_sec", "value": head["value"], "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64+f32/f64", "data": "synthetic",
            "config": head["config"], "roofline": head["roofline"], "verify": head["verify"],
        }
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        if "per_rank" in head:
            out["per_rank"] = head["per_rank"]
        # the metric's second half (BASELINE.json: "+ mean ms/read"; the PAF `mt` tag, README.md:204-217 / mapper.cpp:197): per read, on
        # the device (residence of a read: first event taken up -> decided) and in the reference on the host cores
        g = head["config"].get("gpu_ms_per_read") or {}
        cm = (head.get("cpu_baseline") or {}).get("ms_per_read") or {}
        out["ms_per_read"] = {"gpu_mean": g.get("mean"), "gpu_median": g.get("median"), "cpu_mean": cm.get("mean"), "cpu_median": cm.get("median")}
        if placement:
            out["config"]["host_placement_rank0"] = placement
    # secondary blocks.  N = 1: config 5 (realtime), config 3 (chr20) and one GPU's share of config 4 (GRCh38, 250 k reads).
    # N > 1: config 4 as BASELINE.json states it -- the GRCh38 index replicated on every GPU, every rank its own 250 k reads
    # (2 M reads over 8 GPUs), no collective on the data path; timed like the headline (barriers, MAX over ranks)
    if a.secondary is None:
        a.secondary = "" if not (have_gpu and a.workload == "ecoli") else \
            ("realtime:ecoli,chr20,realtime:chr20,grch38" if world == 1 else "grch38")
    if a.secondary:
        sec = {}
        for w in [x for x in a.secondary.split(",") if x]:
            if w.startswith("realtime") and world > 1:
                continue                    # (per-chunk latency of one flow cell: a single-GPU measurement)
            spent = time.time() - T_START
            if dist is not None:            # every rank takes the same decision (the blocks contain barriers)
                t = torch.tensor([spent], dtype=torch.float64, device=dev_name if have_gpu else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                spent = float(t.item())
            if spent > a.budget_s:
                sec[w] = {"skipped": f"time budget: {spent:.0f} s of {a.budget_s:.0f} s gone"}
                continue
            try:
                t0 = time.time()
                if w.startswith("realtime"):
                    # config 5 on the thresholds of an index this run has already built: realtime[:ecoli|chr20|grch38]
                    ref = w.partition(":")[2] or "ecoli"
                    prefix, codes, lens = ensure_index(cache, rank, barrier, ref, dev_name, lib)
                    rix = capi.Index(prefix, device=local_rank, lib=lib)
                    r = realtime_workload(a, rix, prefix, codes, lens, local_rank, ref, 20, 3, cpu_budget_s=0.0 if a.no_cpu_baseline else 20.0)
                    rix.close()
                    del codes
                    r["wall_s_incl_index_build"] = time.time() - t0
                    sec[w] = r
                    log(f"secondary {w}: {r['value']:.0f} ms per round (p95 {r['config']['latency_ms']['p95']:.0f})")
                    continue
                # one untimed step over the WHOLE batch, then --secondary-steps timed steps (median / min / max in config.step_ms).  (Rounds 2-3 warmed up on 8 192 reads: the first launch
                # that touches all of a 120 GB node pool and 48 GB of slots is 15-20 % slower than the next one -- GRCh38, same
                # process: 29.1 s then 23.6 s -- and was the one that got timed.)
                r = run_workload(a, w, a_reads(a, w), a.secondary_steps, 1, rank, world, local_rank, dist, barrier, cache, lib,
                                 dev_name, extras, 0.0 if a.no_cpu_baseline else 60.0)
                r = {k: v for k, v in r.items() if k != "dt"}
                r["unit"] = "reads/s"
                r["n_gpus"] = world
                r["steps"], r["warmup"] = a.secondary_steps, 1
                r["wall_s_incl_index_build"] = time.time() - t0
                sec[w] = r
                if rank == 0:
                    log(f"secondary {w}: {r['value']:.0f} reads/s on {world} GPU(s)")
            except Exception as e:      # a secondary block never takes the headline down with it
                import traceback
                traceback.print_exc()
                sec[w] = {"error": repr(e)[:400]}
        if sec and rank == 0:
            out["secondary"] = sec
    if rank == 0:
        out["bench_wall_s"] = time.time() - T_START
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
