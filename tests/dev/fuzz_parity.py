#!/usr/bin/env python3
"""Dev checker (CPU): randomised parity of the kernel sources under the lanesim emulator against the oracle -- random reads
(length, noise, dwell, off-target share) on the example index and on a small synthetic reference, random parameter sets drawn
around the defaults, random mapper geometry (slots, slice length, wavefronts, tiny pools).  Everything is seeded: a failure
prints the seed that reproduces it.

    python tests/dev/fuzz_parity.py [n_rounds] [first_seed] [rt]        (rt: the chunked path instead of the batch path; wide: the batch path with 128-bit sort keys;
                                                                         t1: the batch path in UNC_ORDER_T1 with path buffers small enough to leak flags;
                                                                         xcd: the batch path on a mapper whose scheduler rings are per XCD -- 512 / 1 024 slots, 16 - 64 wavefronts,
                                                                         slices of 5 - 64 events, ~100 reads per round: every read parked and resumed dozens of times by other
                                                                         wavefronts of its XCD.  Meant for the GPU: UNC_FUZZ_LIB=uncalled_amd/libuncalled_hip.so)"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyoracle as po  # noqa: E402
from tests.helpers import assert_hits_equal, oracle_hits, to_oracle_params  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE, simulate_reads  # noqa: E402
from uncalled_amd import capi  # noqa: E402
from uncalled_amd.build_index import build_from_codes, encode_contigs, read_fasta, synthetic_genome  # noqa: E402

import os
L = capi.load(os.environ.get("UNC_FUZZ_LIB") or (ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"))
G = ROOT / "tests" / "golden"


def refs(tmp):
    ex = G / "example_index" / "example_ref"
    names, _, seqs = read_fasta(str(ex) + ".fa")
    out = [(ex, encode_contigs(seqs)[0], [len(s) for s in seqs])]
    n2, l2, c2 = synthetic_genome(3, 60000, seed=77)
    pre = Path(tmp) / "fz"
    build_from_codes(pre, n2, [""] * 3, l2, c2)
    Path(str(pre) + ".uncl").write_text("default\t-10.07,-4.6,-4.0,-3.6,-3.3,-3.1\t0.3\t115.000\n")
    out.append((pre, c2, l2))
    if os.environ.get("UNC_FUZZ_MID"):
        # UNC_FUZZ_MID=1: ONLY a mid-sized reference with permissive thresholds (tests/parity_cases.py:case_mid_reference) -- events of
        # hundreds of children, so that every round goes through the runs-and-merge sort (and, chunked, the team's own sort) instead
        # of the network the small references mostly take.  Ten times slower per round.
        n3, l3, c3 = synthetic_genome(1, 800000, seed=11)
        pre3 = Path(tmp) / "mid"
        build_from_codes(pre3, n3, [""], l3, c3)
        Path(str(pre3) + ".uncl").write_text("default\t-10.07,-5.5,-5.0,-4.6,-4.3,-4.1\t0.3\t115.000\n")
        return [(pre3, c3, l3)]
    return out


def draw_params(rng):
    p = capi.default_params(L)
    if rng.random() < 0.7:
        p.max_paths = int(rng.choice([60, 150, 400, 1000, 10000]))
        p.max_consec_stay = int(rng.integers(2, 12))
        p.max_rep_copy = int(rng.integers(1, 65))
        p.min_rep_len = int(rng.integers(0, 15))
        p.max_stay_frac = float(rng.uniform(0.2, 0.7))
        p.min_seed_prob = float(rng.uniform(-4.2, -3.0))
        p.max_events = int(rng.choice([200, 600, 2000, 30000]))
        p.min_map_len = int(rng.integers(12, 40))
        p.min_mean_conf = float(rng.uniform(2.0, 8.0))
        p.min_top_conf = float(rng.uniform(1.1, 2.5))
        p.threshold1 = float(rng.uniform(1.2, 1.8))
        p.threshold2 = float(rng.uniform(7.0, 10.0))
        p.peak_height = float(rng.uniform(0.1, 0.4))
    return p


def rt_round(seed, dix, oix, codes, lens):
    """chunked path: random reads over 1-3 channels, random chunk length / max_chunks / parameters, per channel in order"""
    from uncalled_amd.realtime import MapPoolOrd
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 6))
    sim = simulate_reads(codes, lens, n, seed=seed, read_bases=int(rng.integers(300, 1500)), off_target=float(rng.choice([0.0, 0.3, 1.0])),
                         dwell_mean=float(rng.uniform(6.0, 12.0)), noise_sd=float(rng.uniform(0.5, 3.0)))
    p = draw_params(rng)
    chunk_len = int(rng.choice([1000, 2000, 4000, 8000]))
    p.chunk_time = chunk_len / p.sample_rate
    max_chunks = int(rng.choice([1, 2, 3, 1000000]))
    p.max_chunks = max_chunks
    n_ch = int(rng.integers(1, 4))
    pool = MapPoolOrd(dix, n_channels=n_ch, params=p)
    oms = [po.Mapper(oix, to_oracle_params(p)) for _ in range(n_ch)]
    for om in oms:
        om.set_max_chunks(max_chunks)
    want, got, used, ended = {}, {}, {}, {}
    off = sim["offsets"]
    cal = (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    for i in range(n):
        raw = sim["signal"][int(off[i]):int(off[i + 1])]
        pool.add_read(i % n_ch, i, raw, cal, key=i)
        want[i], used[i] = oms[i % n_ch].chunk_read(po.calibrate(raw, *cal), chunk_len)
        ended[i] = oms[i % n_ch].rt_ended()
    rounds = 0
    while pool.running():
        for key, r in pool.update():
            got[key] = r
        rounds += 1
        assert rounds < 5000
    names = dix.seq_names()
    for i in range(n):
        h, o = got[i]["hit"], want[i]
        assert int(h["status"]) == 0, (i, int(h["status"]))
        assert capi.hit_paf_cols(h, names) == po.hit_paf_cols(o, oix.ref_names()), (i, capi.hit_paf_cols(h, names), po.hit_paf_cols(o, oix.ref_names()))
        for f in ("event_i", "n_nbr", "n_sa", "n_lf", "notes"):
            assert int(h[f]) == int(o[f]), (i, f, int(h[f]), int(o[f]))
        assert pool.chunks_used[i] == used[i], (i, "chunks", pool.chunks_used[i], used[i])
        assert bool(got[i]["ended"]) == ended[i], (i, "ended", int(got[i]["ended"]), ended[i])
    return n, f"{n_ch} channels, chunks of {chunk_len}, max_chunks {max_chunks}, max_paths {p.max_paths}"


def t1_round(seed, dix, oix, codes, lens):
    """`uncalled map -t 1`: ONE mapper, the reads back to back in random batch splits, path buffers small enough that reads leave
    sources_added_ flags behind; against the oracle's shared Mapper.  Counts the reads the carry-over changed."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 9))
    sim = simulate_reads(codes, lens, n, seed=seed, read_bases=int(rng.integers(300, 1200)), off_target=float(rng.choice([0.0, 0.3])),
                         dwell_mean=float(rng.uniform(6.0, 12.0)), noise_sd=float(rng.uniform(0.5, 3.0)))
    p = draw_params(rng)
    p.max_paths = int(rng.choice([40, 60, 97, 130, 200, 400]))
    n_waves = int(rng.integers(1, 3))
    kw = dict(n_waves=n_waves, n_slots=n_waves * int(rng.integers(1, 4)), slice_events=int(rng.choice([0, 13, 64, 1024])))
    raw, off = sim["signal"], sim["offsets"]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    want = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=False)
    indep = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=True)
    changed = sum(1 for i in range(n) if any(int(want[i][f]) != int(indep[i][f]) for f in ("event_i", "n_nbr", "n_sa", "mapped", "rf_st")))
    m = capi.Mapper(dix, params=p, **kw)
    m.set_read_order(capi.ORDER_T1)
    cuts = sorted(set(int(c) for c in rng.integers(1, n, size=int(rng.integers(0, 3)))))
    got, again, lo = [], 0, 0
    for hi in cuts + [n]:
        if hi == lo:
            continue
        b_off = (off[lo:hi + 1] - off[lo]).astype(np.uint64)
        got.append(m.map_batch(raw[int(off[lo]):int(off[hi])], b_off, cal[lo:hi]))
        again += m.last_carry_over()[0]
        lo = hi
    assert_hits_equal(np.concatenate(got), want, f"seed {seed}, -t 1 order")
    return n, f"{kw}, max_paths {p.max_paths}, batches cut at {cuts}, carry-over changes {changed} reads, {again} mapped again"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rt = len(sys.argv) > 3 and sys.argv[3] == "rt"
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix="unc_fuzz_") as tmp:
        R = refs(tmp)
        idx = [(capi.Index(pre, lib=L), po.Index(pre), codes, lens) for pre, codes, lens in R]
        if len(sys.argv) > 3 and sys.argv[3] == "wide":      # the 128-bit-key instantiation (human-sized references) on the small ones
            os.environ["UNC_WIDE_KEYS"] = "1"
            idx = [(capi.Index(pre, lib=L), oix, codes, lens) for (pre, codes, lens), (_, oix, _, _) in zip(R, idx)]
            del os.environ["UNC_WIDE_KEYS"]
        n_reads_total = 0
        for k in range(rounds):
            seed = seed0 + k
            rng = np.random.default_rng(seed)
            dix, oix, codes, lens = idx[int(rng.integers(0, len(idx)))]
            t1 = len(sys.argv) > 3 and sys.argv[3] == "t1"
            if rt or t1:
                try:
                    n, what = (rt_round if rt else t1_round)(seed, dix, oix, codes, lens)
                except Exception as e:
                    print(f"FAILED ({'chunked path' if rt else '-t 1 order'}) at seed {seed}: {e!r}"[:700], flush=True)
                    return 1
                n_reads_total += n
                print(f"seed {seed}: {n} reads ok ({what}) [{time.time() - t0:.0f} s]", flush=True)
                continue
            xcd = len(sys.argv) > 3 and sys.argv[3] == "xcd"
            n = int(rng.integers(2, 6)) if not xcd else int(rng.integers(*[int(x) for x in os.environ.get("UNC_FUZZ_XCD_READS", "80,141").split(",")]))
            sim = simulate_reads(codes, lens, n, seed=seed, read_bases=int(rng.integers(300, 1800 if not xcd else 1000)), off_target=float(rng.choice([0.0, 0.3, 1.0])),
                                 dwell_mean=float(rng.uniform(6.0, 12.0)), noise_sd=float(rng.uniform(0.5, 3.0)))
            p = draw_params(rng)
            n_waves = int(rng.integers(1, 3))
            kw = dict(n_waves=n_waves, n_slots=n_waves * int(rng.integers(1, 4)), slice_events=int(rng.choice([0, 13, 64, 1024])))
            if xcd:
                kw = dict(n_waves=int(rng.choice([16, 32, 64])), n_slots=int(rng.choice([512, 1024])), slice_events=int(rng.choice([5, 13, 29, 64])), pool_chunks=2048)
            if rng.random() < 0.3 and not xcd:
                kw["pool_chunks"] = int(rng.integers(1, 4))
            if rng.random() < 0.2 and not xcd:
                kw["max_clusters"] = int(rng.choice([8, 64]))
            cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
            try:
                tight = "pool_chunks" in kw or "max_clusters" in kw
                mp = capi.Mapper(dix, params=p, **kw)
                if xcd:
                    kw["rings"] = mp.sched_parts()
                hits = mp.map_batch(sim["signal"], sim["offsets"], cal, allow_overflow=tight)
                want = oracle_hits(oix, sim["signal"], sim["offsets"], cal, to_oracle_params(p), fresh_mapper_per_read=True)
                loud = np.flatnonzero(hits["status"])
                if loud.size:
                    # a pool or an allowance too small for these reads even when a read has it to itself (the mid-sized reference: tens
                    # of thousands of seeds per read): reported per read, the OTHER reads are right, and with room every read is
                    ok = np.flatnonzero(hits["status"] == 0)
                    assert_hits_equal(hits[ok], want[ok], f"seed {seed}, reads beside the overflowed ones")
                    kw = {k: v for k, v in kw.items() if k not in ("pool_chunks", "max_clusters")}
                    hits = capi.Mapper(dix, params=p, **{k: v for k, v in kw.items() if k != "rings"}).map_batch(sim["signal"], sim["offsets"], cal)
                    kw["loud_overflows_first"] = int(loud.size)
                assert_hits_equal(hits, want, f"seed {seed}")
            except Exception as e:
                print(f"FAILED at seed {seed}: {kw} max_paths={p.max_paths} max_events={p.max_events}: {e!r}"[:600], flush=True)
                return 1
            n_reads_total += n
            print(f"seed {seed}: {n} reads ok ({kw}, max_paths {p.max_paths}, mapped {int(hits['mapped'].sum())}) [{time.time() - t0:.0f} s]", flush=True)
    print(f"{rounds} rounds, {n_reads_total} reads: device == oracle")
    try:     # (emulator build only) how often an event's walk was done again on 128-bit keys (k_map.hip: phase_S_wide_redo)
        import ctypes
        print("events walked again on wide keys:", ctypes.c_ulonglong.in_dll(L, "unc_sim_wide_redo_count").value)
    except ValueError:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
