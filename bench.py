#!/usr/bin/env python3
"""bench.py -- reads mapped/sec of the MI355X hot path (BASELINE.json metric) on the E. coli configuration.

One step = one pass of the whole hot path (event detection -> normalisation -> match -> FM path forest -> seed
clustering -> PAF coordinates) over one batch of synthetic r9.4.1 reads that is already resident in HBM.
`--gpus N` is launched by the driver as N ranks (torch.distributed.run); reads shard across ranks (index
replicated per GPU, no data-path collective), per-GPU work fixed => weak scaling.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how roofline / cpu_baseline are derived).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)


def ensure_index(cache, rank, world, barrier, workload="ecoli", device=None):
    """SURVEY 8(d) `ecoli_syn`: 1 contig, 4 641 652 bp, i.i.d. ACGT, GC 0.508, seed 1 -> BWA-format index.
    `chr20`: 64 444 167 bp, seed 2, 30 % in N-runs (suffix array built on the GPU)."""
    from uncalled_amd.build_index import build_from_codes, masked_synthetic_genome, synthetic_genome
    if workload == "chr20":
        prefix = cache / "chr20_syn"
        names, lens, codes, holes, n_ambs = masked_synthetic_genome(1, 64444167, seed=2, name="chr20_syn")
    elif workload == "hs400":
        # an eighth of `grch38_syn`: 8 contigs, 400 Mbp, seed 3, 30 % masked -- the largest reference the
        # GPU suffix-array builder takes (seq_len < 2^31); BWT + Occ 400 MB, i.e. past the 256 MiB Infinity Cache
        prefix = cache / "hs400_syn"
        names, lens, codes, holes, n_ambs = masked_synthetic_genome(8, 400000000, seed=3, name="hs400_syn")
    elif workload == "grch38":
        # SURVEY 8(d) `grch38_syn`: 24 contigs, 3.1 Gbp, seed 3, 30 % masked -- seq_len 6.2 G, past the 2^31 limit of the
        # other builders: uncalled_amd/build_index_big.py (chunked suffix sort on the GPU).  NOT YET RUN on the GPU (round 1).
        from uncalled_amd.build_index_big import big_masked_genome
        prefix = cache / "grch38_syn"
        names, lens, codes, holes, n_ambs = big_masked_genome(24, 3100000000, seed=3, name="grch38_syn")
    else:
        prefix = cache / "ecoli_syn"
        names, lens, codes = synthetic_genome(1, 4641652, seed=1)
        holes, n_ambs = (), None
    if rank == 0 and not (Path(str(prefix) + ".sa").exists() and Path(str(prefix) + ".uncl").exists()):
        cache.mkdir(parents=True, exist_ok=True)
        if workload == "grch38":
            from uncalled_amd.build_index_big import build_from_codes_big
            build_from_codes_big(prefix, names, [""] * len(names), lens, codes, holes, n_ambs, device=device, verbose=True)
        else:
            build_from_codes(prefix, names, [""] * len(names), lens, codes, holes, n_ambs,
                             sa_device=device if workload in ("chr20", "hs400") else None)
        # `uncalled index`: thresholds for THIS reference (self-alignment on the GPU + IndexParameterizer, preset
        # "default" = tgt_speed 115, scripts/uncalled:58); build_from_codes left the example's vector as a placeholder
        from uncalled_amd import capi
        from uncalled_amd.index_params import parameterize
        tmp_ix = capi.Index(prefix, device=int(str(device).split(":")[-1]) if device else 0)
        parameterize(tmp_ix, prefix)
        tmp_ix.close()
    barrier()
    return prefix, codes, lens


def mapper_slots(mapper):
    """resident wavefronts / reads in flight / slice length / larger seed-cluster buffers of the k_map scheduler"""
    try:
        return mapper.geometry()
    except Exception as e:      # never let a diagnostic field break the bench line
        return {"error": repr(e)}


def algorithmic_bytes(hits, offsets):
    """per read 2*S + 128*N_nbr + 64*N_lf + 8*N_sa + 64; split per kernel in DESIGN.md."""
    S = (offsets[1:] - offsets[:-1]).astype(np.float64)
    ev_bytes = 2.0 * S + 4.0 * hits["n_events"] + 24.0
    map_bytes = 4.0 * hits["event_i"] + 128.0 * hits["n_nbr"] + 64.0 * hits["n_lf"] + 8.0 * hits["n_sa"] + 64.0
    return float(ev_bytes.sum()), float(map_bytes.sum())


def cpu_baseline(prefix, sim_signal_host, offsets, calib, hits_gpu, seconds_target=20.0):
    """Times the CPU path on the host cores over a bounded sample of the same reads (rank 0, N=1 only) and checks
    the GPU's PAF columns against it.  Prefers oracle/_ref (the reference's own object code) when it travelled."""
    from oracle import pyoracle as po
    from oracle import pyref
    cores = os.cpu_count() or 1
    kind = "reference" if pyref.available() else "port"
    n_avail = offsets.size - 1
    # measured on the GPU box's 256 oversubscribed host threads: ~3 thread-seconds per read with this reference's
    # thresholds; size the sample for about seconds_target of wall time
    n = int(min(n_avail, max(cores * 2, seconds_target * cores / 3.0)))
    off = offsets[:n + 1]
    raw = sim_signal_host[:int(off[-1])]
    sig = po.calibrate(raw, float(calib["range"][0]), float(calib["offset"][0]), float(calib["digitisation"][0]))
    oix = po.Index(prefix)
    names = oix.ref_names()
    if kind == "reference":
        pyref.init(prefix)
        hits, secs = pyref.map_batch(sig, off, cores)
        cpu_cols = [h.paf_cols() for h in hits]
    else:
        hits, secs = po.map_batch(oix, sig, off, cores)
        cpu_cols = [po.hit_paf_cols(h, names) for h in hits]
    from uncalled_amd import capi
    mism = sum(1 for i in range(n) if capi.hit_paf_cols(hits_gpu[i], names) != cpu_cols[i])
    return dict(value=n / secs, unit="reads/s", cores=cores, kind=kind,
                sample=f"first {n} reads of the batch, {cores} threads, tight new_read->map_read loop (BASELINE.md B1)",
                seconds=secs, paf_mismatches_vs_gpu=mism)


def realtime_workload(a, ix, codes, lens, local_rank):
    """BASELINE config 5: 512 channels x 4000-sample chunks, deterministic MAP_ORD-style scheduling; one step = one
    chunk round (every channel hands over its next chunk, all chunks are mapped completely).  Latency is per round."""
    import torch
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from tools.simulate_reads_torch import simulate_reads_torch
    from uncalled_amd import capi
    n_ch, reads_per_ch = a.channels, 6
    sim = simulate_reads_torch(codes, lens, n_ch * reads_per_ch, seed=777, device=f"cuda:{local_rank}")
    off = sim["offsets"].astype(np.int64)
    rt = capi.Realtime(ix, n_channels=n_ch)
    chunk_len = 4000
    cur_read = [0] * n_ch
    cur_chunk = [0] * n_ch
    raw_ptr = sim["signal"].data_ptr()
    lat, ms_ev, ms_map, n_chunks_done, finished = [], [], [], 0, 0
    total_rounds = a.warmup + a.steps
    for rnd in range(total_rounds):
        ch = np.zeros(n_ch, dtype=capi.RT_CHUNK)
        k = 0
        for c in range(n_ch):
            if cur_read[c] >= reads_per_ch:
                cur_read[c] = 0     # replay the channel's reads: the channel never idles
            r = c * reads_per_ch + cur_read[c]
            n = int(off[r + 1] - off[r])
            st = min(cur_chunk[c] * chunk_len, n)
            ln = min(chunk_len, n - st)
            fl = (capi.RT_FIRST if cur_chunk[c] == 0 else 0) | (capi.RT_LAST if st + ln >= n else 0)
            ch[k]["channel"], ch[k]["read_number"], ch[k]["flags"], ch[k]["n_samples"], ch[k]["offset"] = c, cur_read[c] + rnd * 1000, fl, ln, off[r] + st
            ch[k]["calib"]["range"], ch[k]["calib"]["offset"], ch[k]["calib"]["digitisation"] = CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION
            k += 1
        # read numbers must stay constant within a read: derive from (channel replay count, read index)
        for j in range(k):
            c = int(ch[j]["channel"])
            ch[j]["read_number"] = cur_read[c]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = rt.process_chunks(ch[:k], raw_ptr=raw_ptr)
        dt = time.perf_counter() - t0
        e, m = rt.last_timing()
        if rnd >= a.warmup:
            lat.append(dt * 1e3); ms_ev.append(e); ms_map.append(m); n_chunks_done += k
        for j in range(k):
            c = int(ch[j]["channel"])
            if res[j]["state"] == capi.RT_MAPPING:
                cur_chunk[c] += 1
            else:
                cur_chunk[c] = 0
                cur_read[c] += 1
                finished += 1
    lat = np.array(lat)
    return {"metric": "chunk_round_latency_ms", "value": float(lat.mean()), "unit": "ms", "n_gpus": 1, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": float(lat.mean()), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64+f32/f64", "data": "synthetic",
            "config": {"workload": f"realtime: {n_ch} channels x {chunk_len}-sample chunks (1 s of signal each), E. coli synthetic ref, "
                                   "MAP_ORD-style deterministic scheduling, raw signal resident in HBM",
                       "latency_ms": {"mean": float(lat.mean()), "p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)),
                                      "max": float(lat.max())},
                       "chunks_per_sec": n_chunks_done / (lat.sum() * 1e-3), "reads_finished": finished,
                       "kernel_ms": {"k_rt_events": float(np.mean(ms_ev)), "k_map": float(np.mean(ms_map))},
                       "sla": "a chunk is 1000 ms of signal; the round must finish well inside that"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("UNC_BENCH_READS", 50000)),
                    help="reads per GPU per step (config: E. coli 4.6 Mb ref, 50k synthetic r9.4.1 reads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-big", type=int, default=0, help="larger seed-cluster buffers (0 = library default)")
    ap.add_argument("--big-clusters", type=int, default=0, help="clusters per larger buffer (0 = library default)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the extra untimed pass that collects phase cycle shares")
    ap.add_argument("--workload", choices=["ecoli", "chr20", "hs400", "grch38", "realtime"], default="ecoli")
    ap.add_argument("--channels", type=int, default=512)
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from tools.simulate_reads_torch import simulate_reads_torch
    from uncalled_amd import capi

    cache = Path(os.environ.get("UNC_BENCH_CACHE", "/tmp/uncalled_amd_bench"))
    prefix, codes, lens = ensure_index(cache, rank, world, barrier, a.workload if a.workload in ("chr20", "hs400", "grch38") else "ecoli",
                                       f"cuda:{local_rank}")
    ix = capi.Index(prefix, device=local_rank)
    if a.workload == "realtime":
        out = realtime_workload(a, ix, codes, lens, local_rank)
        if rank == 0:
            print(json.dumps(out))
        return
    mapper = capi.Mapper(ix, n_big=a.n_big, big_clusters=a.big_clusters)
    # this rank's shard of the read set: reads are independent units, sharded by rank with distinct seeds
    sim = simulate_reads_torch(codes, lens, a.reads, seed=42 + rank, device=f"cuda:{local_rank}")
    offsets = sim["offsets"]
    calib = capi.make_calib(a.reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    raw_ptr = sim["signal"].data_ptr()
    stream = torch.cuda.current_stream().cuda_stream

    hits = None
    for _ in range(a.warmup):
        hits = mapper.map_batch_device(raw_ptr, offsets, calib, stream=stream)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    ms_ev, ms_map = [], []
    for _ in range(a.steps):
        hits = mapper.map_batch_device(raw_ptr, offsets, calib, stream=stream)
        e, m = mapper.last_timing()
        ms_ev.append(e)
        ms_map.append(m)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    wave_busy = mapper.last_wave_busy()
    remap_n, remap_ms = mapper.last_remap()
    # phase shares come from one extra, untimed pass with the cycle-counting instantiation of k_map
    # the boundary also takes host buffers (unc_map_batch on_device = 0): one extra, untimed step from pageable host
    # memory gives the PCIe-inclusive rate (never `value`)
    pcie = None
    if rank == 0 and world == 1 and not a.no_profile_pass:
        host_raw = sim["signal"].cpu().numpy()
        t1 = time.perf_counter()
        mapper.map_batch(host_raw, offsets, calib)
        pcie = a.reads / (time.perf_counter() - t1)
        del host_raw
    if not a.no_profile_pass:
        mapper.set_profile(True)
        mapper.map_batch_device(raw_ptr, offsets, calib, stream=stream)
        mapper.set_profile(False)
    pc = mapper.last_phase_cycles()
    tot_c = float(sum(pc.values())) or 1.0
    phase_share = {k: round(v / tot_c, 4) for k, v in pc.items()}
    if rank == 0:
        total_reads = a.reads * world * a.steps
        ev_bytes, map_bytes = algorithmic_bytes(hits, offsets)
        map_ms = float(np.mean(ms_map))
        achieved = map_bytes / (map_ms * 1e-3) / 1e9
        # HBM traffic of k_map from the committed rocprofv3 PMC passes of this same command/config
        # (FETCH_SIZE and WRITE_SIZE cannot be collected from inside this process), scaled per read
        traffic = None
        pmc = ROOT / "profiles" / "r01_pmc_k_map.json"
        if pmc.exists():
            traffic = json.loads(pmc.read_text())["hbm_bytes_per_read"] * a.reads
        out = {
            "metric": "reads_mapped_per_sec", "value": total_reads / dt, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64+f32/f64",
            "data": "synthetic",
            "config": {"workload": ("repeat-masked chr20-sized synthetic ref (chr20_syn 64.4 Mb, seed 2, 30% N-runs)" if a.workload == "chr20"
                                    else "one eighth of a masked GRCh38-sized synthetic ref (hs400_syn: 8 contigs, 400 Mb, seed 3, 30% N-runs)"
                                    if a.workload == "hs400" else "masked GRCh38-sized synthetic ref (grch38_syn: 24 contigs, 3.1 Gb, seed 3, 30% N-runs)"
                                    if a.workload == "grch38" else "E. coli 4.6 Mb synthetic ref (ecoli_syn seed 1)") +
                                   ", synthetic r9.4.1 reads (3600 bases ~ 32k samples, 10% off-target), all reference defaults",
                       "reads_per_gpu_per_step": a.reads, "parallelism": f"reads sharded over {world} GPU(s), index replicated",
                       "mean_ms_per_read_amortised": 1e3 * dt / (a.reads * a.steps),
                       "mapped_fraction": float(hits["mapped"].mean()),
                       "mean_events_per_read": float(hits["event_i"].mean()),
                       "kernel_ms": {"k_events": float(np.mean(ms_ev)), "k_map": map_ms},
                       "k_map_phase_cycle_share": phase_share,
                       "k_map_phase_cycle_share_source": "extra untimed pass, profiling instantiation of k_map",
                       "k_map_wave_busy": round(wave_busy, 4),
                       "pcie_inclusive_reads_per_sec": pcie,
                       "remapped_reads": {"n": remap_n, "ms": round(remap_ms, 1),
                                          "note": "reads whose seed-cluster set outgrew its slot, mapped again with 16x the room (inside the step)"},
                       "reads_in_flight": mapper_slots(mapper)},
            "roofline": {"bound": "hbm", "kernel": "k_map", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_k_map.json (FETCH_SIZE+WRITE_SIZE per read x reads)",
                         "algorithmic_bytes_per_launch": map_bytes, "launch_ms": map_ms,
                         "whole_path_bytes_per_step": ev_bytes + map_bytes},
        }
        if world == 1 and not a.no_cpu_baseline:
            host_sig = sim["signal"][:int(offsets[min(a.reads, 4096)])].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(prefix, host_sig, offsets[:min(a.reads, 4096) + 1], calib, hits)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
