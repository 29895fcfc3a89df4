"""Parity cases shared by the lanesim (CPU emulator, container) and gpu (real MI355X) test modules: the HIP path
-- through the C ABI -- against the oracle restatement and the committed reference goldens."""
from pathlib import Path

import numpy as np
import pytest

from tests.helpers import assert_hits_equal, oracle_hits, to_oracle_params
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from uncalled_amd import capi

_cache = {}
GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


def _index(lib, example):
    key = (id(lib), str(example["prefix"]))
    if key not in _cache:
        _cache[key] = capi.Index(example["prefix"], lib=lib)
    return _cache[key]


def case_index_tables(lib, oracle_lib, example, goldens):
    dev_index = _index(lib, example)
    oix = oracle_lib.Index(example["prefix"])
    assert np.array_equal(dev_index.kmer_ranges(), oix.kmer_ranges())
    assert np.array_equal(dev_index.thresholds(), oix.thresholds())
    for a, b in zip(dev_index.model_tables(), oracle_lib.model_tables()):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def case_fm_primitives(lib, oracle_lib, example, goldens):
    dev_index = _index(lib, example)
    oix = oracle_lib.Index(example["prefix"])
    rng = np.random.default_rng(1)
    kr = oix.kmer_ranges()
    s, e, b = [], [], []
    for k in rng.integers(0, 1024, 300):
        if kr[k, 0] <= kr[k, 1]:
            lo = int(rng.integers(kr[k, 0], kr[k, 1] + 1))
            hi = int(rng.integers(lo, kr[k, 1] + 1))
            s.append(lo); e.append(hi); b.append(int(rng.integers(0, 4)))
    # edge rows: first row, the sentinel row's neighbours, last row
    for row in (1, 2, int(oix.size) - 1, int(oix.size)):
        s.append(row); e.append(row); b.append(int(rng.integers(0, 4)))
    os_, oe = dev_index.get_neighbor(s, e, b)
    for x, y, z, p, q in zip(s, e, b, os_, oe):
        assert oix.get_neighbor(x, y, z) == (int(p), int(q))
    rows = np.concatenate((rng.integers(0, oix.size + 1, 400), [0, 1, oix.size]))
    assert np.array_equal(dev_index.sa(rows), np.array([oix.sa(int(r)) for r in rows], dtype=np.uint64))


def case_events_and_normaliser(lib, oracle_lib, example, goldens):
    dev_index = _index(lib, example)
    m = capi.Mapper(dev_index, n_slots=1)
    raw = example["signal"]
    off = np.array([0, raw.size], dtype=np.uint64)
    cal = capi.make_calib(1, example["range"], example["offset"], example["digitisation"])
    means, moff, info = m.detect_events(raw, off, cal)
    assert np.array_equal(means, goldens["ex_events"]["mean"])          # north star: 1e-5; achieved: bit-exact
    assert info["total_events"][0] == int(goldens["ex_total_events"])
    assert np.float32(info["len_sum"][0] / info["total_events"][0]) == goldens["ex_mean_event_len"]
    assert info["scale"][0] == goldens["ex_scale"] and info["shift"][0] == goldens["ex_shift"]
    lv = goldens["ex_levels"]
    assert np.array_equal(dev_index.match_probs(lv[:64]), goldens["ex_probs"])


def case_radix_sort(lib, big=False):
    rng = np.random.default_rng(1)
    sizes = [(1, 64), (63, 8), (64, 16), (2048, 24), (2049, 64), (5000, 62), (70001, 40)] + ([(3_000_001, 63), (1 << 22, 33)] if big else [(300000, 33)])
    for n, bits in sizes:
        keys = rng.integers(0, 2 ** min(bits, 63), size=n, dtype=np.uint64)
        if bits == 64:
            keys = keys * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        if n > 100:
            keys[::7] = keys[3]          # ties: a seventh of the keys are equal
        want = np.argsort(keys, kind="stable").astype(np.uint64)
        sk, perm = capi.sort_pairs(keys, None, bits, lib=lib)
        assert np.array_equal(perm, want) and np.array_equal(sk, keys[want]), (n, bits)
        vals = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)
        sk2, sv = capi.sort_pairs(keys, vals, bits, lib=lib)
        assert np.array_equal(sk2, sk) and np.array_equal(sv, vals[want]), (n, bits)


def case_events_sweep_reads(lib, oracle_lib, example):
    """Reads the round-5 parity sweeps (10 240 reads each of the chr20 and GRCh38 bench batches against the reference's object code,
    tests/dev/parity_sweep.py) found the device WRONG on: one event too many, because the t-statistic's square root was hipcc's
    __fsqrt_rn = the bare v_sqrt_f32 (1 ulp) and not the correctly rounded sqrtss of the reference (event_detector.cpp:218).  One read
    in seven thousand; the emulator (sqrtf on the host) never saw it.  The raw signals are committed (tests/golden/sweep_reads_r05.npz,
    dumped on the GPU box by tests/dev/dump_reads.py); every kept event mean and both counters must be the oracle's, at every alignment
    of the read inside its batch.  (The oracle's detector is pinned on the reference's own: tests/test_oracle.py.)"""
    d = np.load(GOLDEN_DIR / "sweep_reads_r05.npz")
    dev_index = _index(lib, example)
    m = capi.Mapper(dev_index, n_slots=4)
    for r in (37043, 66882, 61770):
        raw = d[f"raw_{r}"]
        ev, mel, tot = oracle_lib.detect_events(oracle_lib.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))[:3]
        for pad in (0, 1, int(8 - int(d[f"offmod_{r}"])) % 8 + 8):
            full = np.concatenate((np.full(pad, 500, np.int16), raw))
            off = np.array([0, pad, pad + raw.size], dtype=np.uint64) if pad else np.array([0, raw.size], dtype=np.uint64)
            means, moff, info = m.detect_events(full, off, capi.make_calib(off.size - 1, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
            k = off.size - 2
            assert info["n_events"][k] == len(ev) and info["total_events"][k] == tot, (r, pad, int(info["n_events"][k]), len(ev))
            assert np.array_equal(means[int(moff[k]):int(moff[k + 1])], ev["mean"].astype(np.float32)), (r, pad)
            assert np.float32(info["len_sum"][k] / info["total_events"][k]) == np.float32(mel), (r, pad)


def case_events_edge_cases(lib, oracle_lib, example, goldens):
    """empty read, reads shorter than the detector windows, a flat (zero-variance) read, negative samples."""
    dev_index = _index(lib, example)
    po = oracle_lib
    rng = np.random.default_rng(5)
    reads = [np.zeros(0, np.int16), np.array([500], np.int16), rng.integers(300, 700, 12).astype(np.int16),
             np.full(400, 512, np.int16), rng.integers(-200, 900, 700).astype(np.int16),
             np.repeat(rng.integers(350, 650, 60), 9).astype(np.int16)]
    # k_events takes 8 samples per step from the first 16-byte boundary past sample 16 on: every head / tail length and
    # every alignment of a read inside the batch (step signals, so that events fall into heads and tails as well)
    for ln in list(range(13, 42)) + [63, 64, 65, 127, 129, 1000, 1003]:
        reads.append(np.repeat(rng.integers(330, 680, ln // 5 + 1), 5)[:ln].astype(np.int16) + rng.integers(-6, 7, ln).astype(np.int16))
    raw = np.concatenate(reads)
    off = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint64)
    cal = capi.make_calib(len(reads), CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    m = capi.Mapper(dev_index, n_slots=1)
    means, moff, info = m.detect_events(raw, off, cal)
    for i, r in enumerate(reads):
        ev, mel, tot = po.detect_events(po.calibrate(r, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
        assert info["n_events"][i] == len(ev) and info["total_events"][i] == tot, i
        assert np.array_equal(means[int(moff[i]):int(moff[i + 1])], ev["mean"]), i
        if len(ev):
            lv, sc, sh = po.normalize(ev["mean"])
            assert (np.float32(sc) == info["scale"][i] or (np.isnan(sc) and np.isnan(info["scale"][i]))), i
            assert (np.float32(sh) == info["shift"][i] or (np.isnan(sh) and np.isnan(info["shift"][i]))), i


def case_example_read_full_path(lib, oracle_lib, example, goldens):
    dev_index = _index(lib, example)
    m = capi.Mapper(dev_index, n_slots=2)
    raw = example["signal"]
    off = np.array([0, raw.size], dtype=np.uint64)
    cal = capi.make_calib(1, example["range"], example["offset"], example["digitisation"])
    hits = m.map_batch(raw, off, cal)
    oix = oracle_lib.Index(example["prefix"])
    assert_hits_equal(hits, oracle_hits(oix, raw, off, cal), "example")
    assert capi.hit_paf_cols(hits[0], dev_index.seq_names()) == \
        (106, 73, 106, "-", "Escherichia_coli_chromosome:2400000-2410000", 10000, 6938, 6976, 38, 39, 255)


def case_synthetic_batch(lib, oracle_lib, example, goldens, max_paths, n_reads):
    """Batch through the persistent read queue (more reads than slots); small max_paths exercises the buffer
    cut-off semantics of mapper.cpp:480-482,507-509,521-523,544,576,605-624 incl. stale sources_added_ flags."""
    dev_index = _index(lib, example)
    p = capi.default_params(dev_index.L)
    p.max_paths = max_paths
    m = capi.Mapper(dev_index, params=p, n_slots=3)
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    hits = m.map_batch(raw, off, cal)
    oix = oracle_lib.Index(example["prefix"])
    want = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=True)
    assert_hits_equal(hits, want, f"max_paths={max_paths}")
    if max_paths == 10000:
        f = {str(n): j for j, n in enumerate(goldens["hit_fields"])}
        for i in range(n_reads):   # and against the reference's own answers
            for name in ("mapped", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "matches", "event_i", "n_nbr", "n_lf"):
                assert int(hits[i][name]) == int(goldens["sim_hits"][i][f[name]]), (i, name)


def case_read_order_t1(lib, oracle_lib, example, goldens, max_paths, n_reads, split):
    """UNC_ORDER_T1 = `uncalled map -t 1`: ONE Mapper maps the reads back to back (the oracle with a shared mapper; pinned against the
    reference's own Mapper in test_oracle.py), sources_added_ travelling from read to read -- within a batch and, here, across the
    two batches the reads are handed over in.  A small max_paths makes nearly every read leave flags behind; the case also checks
    that this matters (the independent order gives different answers on the same reads), so that it cannot pass vacuously."""
    dev_index = _index(lib, example)
    p = capi.default_params(dev_index.L)
    p.max_paths = max_paths
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    oix = oracle_lib.Index(example["prefix"])
    want = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=False)
    indep = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=True)
    differ = [i for i in range(n_reads) if any(int(want[i][f]) != int(indep[i][f]) for f in ("event_i", "n_nbr", "n_sa", "mapped", "rf_st"))]
    assert differ, "the carry-over changes nothing on these reads: the case proves nothing"
    m = capi.Mapper(dev_index, params=p, n_slots=3)
    m.set_read_order(capi.ORDER_T1)
    a = m.map_batch(raw[:int(off[split])], off[:split + 1].copy(), cal[:split])
    n_a = m.last_carry_over()[0]
    b_off = (off[split:] - off[split]).astype(np.uint64)
    b = m.map_batch(raw[int(off[split]):], b_off, cal[split:])
    n_b = m.last_carry_over()[0]
    assert_hits_equal(np.concatenate([a, b]), want, f"-t 1 order, max_paths={max_paths}")
    assert n_a + n_b > 0
    # a new run starts clear: the first read again as a fresh Mapper maps it
    m.set_read_order(capi.ORDER_T1)
    again = m.map_batch(raw[:int(off[1])], off[:2].copy(), cal[:1])
    assert_hits_equal(again, indep[:1], "first read of a new run")
    # ... and the default order is untouched by all this
    m.set_read_order(capi.ORDER_INDEPENDENT)
    assert_hits_equal(m.map_batch(raw, off, cal), indep, "independent order")


def case_trace_matches_oracle_every_event(lib, oracle_lib, example, goldens, dev_index=None):
    """Path buffer (order, ranges, k-mers, prob-sum windows, flags) and the seed-cluster set after every map_next."""
    dev_index = dev_index or _index(lib, example)
    raw = example["signal"][:6000]
    cal = capi.make_calib(1, example["range"], example["offset"], example["digitisation"])
    m = capi.Mapper(dev_index, n_slots=1)
    oix = oracle_lib.Index(example["prefix"])
    om = oracle_lib.Mapper(oix)
    sig = oracle_lib.calibrate(raw, example["range"], example["offset"], example["digitisation"])
    steps = 0
    for (dd, dpaths, dclus, dmm, dls, dnl), (od, oe, opaths, oclus, omm, ols, onl) in zip(m.trace(raw, cal), om.trace(sig)):
        assert dd == od, steps
        v = opaths["length"] > 0          # the device list holds only valid parents, in the same order
        ov = opaths[v]
        assert len(dpaths) == len(ov), steps
        for f in ("fm_start", "fm_end", "kmer", "length", "event_moves", "seed_prob", "consec_stays", "sa_checked"):
            assert np.array_equal(dpaths[f], ov[f]), (steps, f)
        for j in range(len(ov)):
            L = int(ov["length"][j])
            assert np.array_equal(dpaths["prob_sums"][j][:L + 1], ov["prob_sums"][j][:L + 1]), (steps, j)
        assert np.array_equal(dclus, oclus), steps
        assert dls == ols and dnl == onl, steps
        if omm["total_len"]:
            assert dmm == omm, steps
        steps += 1
    assert steps > 100
    dh, oh = m.trace_finish(), om.trace_finish()
    assert_hits_equal([dh], [oh], "trace")


# Parameter sets away from the reference's defaults (mapper.cpp:29-40, event_detector.cpp:17-26, seed_tracker.cpp:28-32): each
# moves a decision the default run hardly ever takes the other way.  tests/test_oracle.py pins the oracle against the live
# reference on every one of them; the device is compared with the oracle.
PARAM_VARIANTS = [
    dict(max_rep_copy=5, min_rep_len=12),                       # repeat seeds: fewer copies, only after 12 moves (mapper.cpp:858-861)
    dict(max_consec_stay=3, max_stay_frac=0.25),                # stay children cut earlier (:470-483), stricter seed validity (:852-854)
    dict(min_seed_prob=-3.2),                                   # fewer seed-valid paths
    dict(max_events=350),                                       # reads that run out of events (:381-405)
    dict(min_map_len=15, min_mean_conf=3.0, min_top_conf=1.2),  # earlier, weaker success (seed_tracker.cpp:129-143)
    dict(threshold1=1.7, threshold2=8.0, peak_height=0.35, min_mean=55.0, max_mean=130.0),   # another segmentation, events dropped by mean
    dict(max_paths=150, max_rep_copy=64, max_consec_stay=12),   # the max_paths cut-off with long stays and the largest repeat copy
]


def variant_params(base, overrides):
    for k, v in overrides.items():
        setattr(base, k, v)
    return base


def case_parameter_variants(lib, oracle_lib, example, goldens, n=6):
    dev_index = _index(lib, example)
    oix = oracle_lib.Index(example["prefix"])
    off = goldens["sim_offsets"][:n + 1].copy()
    raw = goldens["sim_signal"][:int(off[n])]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    for ov in PARAM_VARIANTS:
        p = variant_params(capi.default_params(lib), ov)
        hits = capi.Mapper(dev_index, params=p, n_slots=3).map_batch(raw, off, cal)
        want = oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=True)
        assert_hits_equal(hits, want, str(ov))


def case_narrow_buckets(lib, oracle_lib, example, goldens, monkeypatch, shift=4):
    """The seed-cluster grid with buckets of 2^shift rows instead of 2^12: on the 20 k-row example index a seed's window then
    spans hundreds of buckets (the lane that adds a seed walks them one after the other), clusters move from bucket to bucket as
    they grow, and every bucket holds a cluster or none -- per event against the oracle's set, then a batch."""
    monkeypatch.setenv("UNC_BUCKET_SHIFT", str(shift))
    ix = capi.Index(example["prefix"], lib=lib)
    monkeypatch.delenv("UNC_BUCKET_SHIFT", raising=False)
    case_trace_matches_oracle_every_event(lib, oracle_lib, example, goldens, dev_index=ix)
    oix = oracle_lib.Index(example["prefix"])
    n = 8
    off = goldens["sim_offsets"][:n + 1].copy()
    raw = goldens["sim_signal"][:int(off[n])]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    hits = capi.Mapper(ix, n_slots=3).map_batch(raw, off, cal)
    assert_hits_equal(hits, oracle_hits(oix, raw, off, cal), "narrow buckets")


def case_loud_overflows(lib, oracle_lib, example, goldens):
    """Device scratch that is too small is REPORTED per read (unc_hit_t.status) and by the call's return code -- never a silent
    wrong answer: a per-event seed list of one entry (UNC_READ_SEED_OVERFLOW); on the chunked path, chunk lengths that could fill the
    rolling normaliser (the reference's #SKIP branch, mapper.cpp:336-351: out of scope) are refused at creation.  Reads that did not
    overflow still answer as the oracle does."""
    dev_index = _index(lib, example)
    oix = oracle_lib.Index(example["prefix"])
    n = 4
    off = goldens["sim_offsets"][:n + 1].copy()
    raw = goldens["sim_signal"][:int(off[n])]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    m = capi.Mapper(dev_index, n_slots=3, max_seed_paths=1)
    with pytest.raises(RuntimeError):
        m.map_batch(raw, off, cal)                       # the call itself fails ...
    hits = m.map_batch(raw, off, cal, allow_overflow=True)
    want = oracle_hits(oix, raw, off, cal)
    bad = (hits["status"] & 2) != 0
    assert bad.any() and not (hits["status"] & ~np.uint32(2)).any()      # ... and says which reads, and why
    assert not hits["mapped"][bad].any()
    ok = np.flatnonzero(~bad)
    assert_hits_equal(hits[ok], [want[i] for i in ok], "reads without seed overflow")
    # chunked path: the reference's #SKIP branch (a ring of 6000 unread events, mapper.cpp:336-351) is out of scope, formally: chunks
    # long enough to reach it are refused when the pool is created; 3 s chunks (12 000 samples, at most 6000 events) are the limit
    p = capi.default_params(lib)
    p.chunk_time = 15.0
    with pytest.raises(capi.UncalledHipError, match="SKIP"):
        capi.Realtime(dev_index, 1, p)
    p.chunk_time = 3.0
    capi.Realtime(dev_index, 1, p).close()
    p.chunk_time = 3.0 + 1.0 / 4000.0
    with pytest.raises(capi.UncalledHipError):
        capi.Realtime(dev_index, 1, p)


# chunked-path sets: (parameter overrides, chunk length in samples)
CHUNK_VARIANTS = [
    (dict(threshold1=1.7, threshold2=8.0, peak_height=0.35, min_mean=55.0, max_mean=130.0), 4000),
    (dict(min_seed_prob=-3.2, max_consec_stay=3, max_stay_frac=0.25), 2000),         # half-second chunks
    (dict(min_map_len=15, min_mean_conf=3.0, min_top_conf=1.2, max_events=350), 1000),
]


def case_chunked_variants(lib, oracle_lib, example, goldens, n_channels=2, n_reads=6):
    """The chunked path on parameter sets away from the defaults and with other chunk lengths (the detector, the event
    profiler, the rolling normaliser and the mapper all see different event streams): device vs oracle, per read."""
    from uncalled_amd.realtime import MapPoolOrd
    po = oracle_lib
    dev_index = _index(lib, example)
    oix = po.Index(example["prefix"])
    off = goldens["sim_offsets"]
    reads = [(example["signal"], (example["range"], example["offset"], example["digitisation"]))]
    for i in range(n_reads - 1):
        reads.append((goldens["sim_signal"][int(off[i]):int(off[i + 1])], (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)))
    names = dev_index.seq_names()
    for ov, chunk_len in CHUNK_VARIANTS:
        p = variant_params(capi.default_params(lib), ov)
        p.chunk_time = chunk_len / p.sample_rate
        pool = MapPoolOrd(dev_index, n_channels=n_channels, params=p)
        assert pool.chunk_len == chunk_len
        oms = [po.Mapper(oix, to_oracle_params(p)) for _ in range(n_channels)]
        want, got, fate = {}, {}, {}
        for i, (raw, cal) in enumerate(reads):
            pool.add_read(i % n_channels, i, raw, cal, key=i)
            want[i], used = oms[i % n_channels].chunk_read(po.calibrate(raw, *cal), chunk_len)
            fate[i] = (used, oms[i % n_channels].rt_ended())
        rounds = 0
        while pool.running():
            for key, r in pool.update():
                got[key] = r
            rounds += 1
            assert rounds < 4000
        for i in range(len(reads)):
            h, o = got[i]["hit"], want[i]
            assert int(h["status"]) == 0
            assert capi.hit_paf_cols(h, names) == po.hit_paf_cols(o, oix.ref_names()), (ov, chunk_len, i)
            for f in ("event_i", "n_nbr", "n_sa", "n_lf", "notes"):
                assert int(h[f]) == int(o[f]), (ov, chunk_len, i, f)
            assert got[i]["state"] == (capi.RT_MAPPED if o["mapped"] else capi.RT_FAILED), (ov, chunk_len, i)
            assert (pool.chunks_used[i], bool(got[i]["ended"])) == fate[i], (ov, chunk_len, i)       # chunks the read was given, Paf::ENDED


def assert_rt_taps_equal(a, ring_a, b, ring_b, what):
    """stage tap of the chunked path (unc_rt_tap_t layout): detector counters, profiler window + queue, rolling normaliser + ring"""
    for f in ("det_t", "det_total_events", "det_len_sum", "norm_n", "norm_wr", "prof_n", "prof_to_mask", "prof_queued"):
        assert a[f] == b[f], (what, f, a[f], b[f])
    for f in ("norm_mean", "norm_varsum", "prof_mean", "prof_varsum"):
        assert np.float64(a[f]).tobytes() == np.float64(b[f]).tobytes(), (what, f, a[f], b[f])
    q = int(a["prof_queued"])
    assert np.array_equal(a["prof_queue"][:q].view(np.uint32), b["prof_queue"][:q].view(np.uint32)), (what, "prof_queue")
    n = int(a["norm_n"])
    assert np.array_equal(ring_a[:n].view(np.uint32), ring_b[:n].view(np.uint32)), (what, "ring")


def case_chunked_stage_tap(lib, oracle_lib, example, goldens, n_reads=8):
    """Below the PAF of the chunked path: after every read of a channel the EventDetector's counters, the EventProfiler's
    25-event window (mean / varsum / mask countdown / queued events) and the rolling 6000-event Normalizer (mean / varsum /
    write position / ring contents) of the device equal the oracle's bit for bit -- a masking or normalisation slip that
    cancelled out in the coordinates would show here.  (tests/test_oracle.py pins the oracle's tap against the reference's.)"""
    from uncalled_amd.realtime import MapPoolOrd
    po = oracle_lib
    dev_index = _index(lib, example)
    oix = po.Index(example["prefix"])
    pool = MapPoolOrd(dev_index, n_channels=1)
    om = po.Mapper(oix)
    off = goldens["sim_offsets"]
    reads = [(example["signal"], (example["range"], example["offset"], example["digitisation"]))]
    for i in range(n_reads - 1):
        reads.append((goldens["sim_signal"][int(off[i]):int(off[i + 1])], (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)))
    for i, (raw, cal) in enumerate(reads):
        pool.add_read(0, i, raw, cal, key=i)
    done = 0
    rounds = 0
    while pool.running():
        for key, r in pool.update():
            raw, cal = reads[key]
            om.chunk_read(po.calibrate(raw, *cal), 4000)
            (ta, ra), (tb, rb) = pool.rt.tap_channel(0), om.rt_tap()
            assert_rt_taps_equal(ta, ra, tb, rb, "read %d" % key)
            done += 1
        rounds += 1
        assert rounds < 1000
    assert done == n_reads and int(tb["norm_n"]) == 6000       # (the ring has wrapped by then)


def fuzz_reference(tmp_path):
    """the second reference of tests/dev/fuzz_parity.py: three contigs of 60 kb, loose thresholds (many children per event)"""
    from uncalled_amd.build_index import build_from_codes, synthetic_genome
    names, lens, codes = synthetic_genome(3, 60000, seed=77)
    prefix = tmp_path / "fz"
    if not (tmp_path / "fz.sa").exists():
        build_from_codes(prefix, names, [""] * 3, lens, codes)
        (tmp_path / "fz.uncl").write_text("default\t-10.07,-4.6,-4.0,-3.6,-3.3,-3.1\t0.3\t115.000\n")
    return prefix, codes, lens


def case_unsorted_stream_past_one_block(lib, oracle_lib, tmp_path):
    """max_paths = 1000 with events of many children of sources: the unsorted key stream holds more than 512 keys and is sorted IN
    PLACE in room for exactly max_paths keys.  Round 3's first form of that sort stored whole 512-key blocks, padding included, i.e.
    past the stream's end into the children's info words (found by tests/dev/fuzz_parity.py, seed 2085: a read unmapped that the
    reference maps; the emulator's bounds checks, UNC_SIM_CHECK, now fail on such a store).  The fuzz case, against the oracle."""
    from tools.simulate_reads import simulate_reads
    prefix, codes, lens = fuzz_reference(tmp_path)
    sim = simulate_reads(codes, lens, 4, seed=2085, read_bases=490, off_target=0.0, dwell_mean=11.29495188046118, noise_sd=1.030862943416349)
    p = capi.default_params(lib)
    for k, v in dict(min_rep_len=1, max_rep_copy=28, max_paths=1000, max_consec_stay=2, max_events=2000, max_stay_frac=0.4174584150314331,
                     min_seed_prob=-4.121715545654297, threshold1=1.4309371709823608, threshold2=9.04516315460205,
                     peak_height=0.31874945759773254, min_map_len=27, min_mean_conf=7.31113862991333, min_top_conf=1.9158369302749634).items():
        setattr(p, k, v)
    cal = capi.make_calib(4, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    ix = capi.Index(prefix, lib=lib)
    hits = capi.Mapper(ix, params=p, n_slots=2, n_waves=2).map_batch(sim["signal"], sim["offsets"], cal)
    want = oracle_hits(oracle_lib.Index(prefix), sim["signal"], sim["offsets"], cal, to_oracle_params(p), fresh_mapper_per_read=True)
    assert_hits_equal(hits, want, "unsorted stream of more than one block")
    assert int(hits["mapped"].sum()) == 2 and (hits["n_nbr"] / np.maximum(hits["event_i"], 1)).min() > 400


CARRY_OVER_PARAMS = dict(max_paths=60, max_rep_copy=2, max_chunks=2)      # with chunks of 8000 samples


def carry_over_data(tmp_path):
    """the reference and reads of tests/dev/fuzz_parity.py's seed 5007 (chunked mode): (index prefix, simulated reads, n)"""
    from tools.simulate_reads import simulate_reads
    prefix, codes, lens = fuzz_reference(tmp_path)
    rng = np.random.default_rng(5007)
    n = int(rng.integers(2, 6))
    sim = simulate_reads(codes, lens, n, seed=5007, read_bases=int(rng.integers(300, 1500)), off_target=float(rng.choice([0.0, 0.3, 1.0])),
                         dwell_mean=float(rng.uniform(6.0, 12.0)), noise_sd=float(rng.uniform(0.5, 3.0)))
    assert n == 4
    return prefix, sim, n


def case_chunked_flags_carry_over(lib, oracle_lib, tmp_path):
    """A channel is one Mapper, and Mapper::new_read does not clear sources_added_ (mapper.cpp:88,612-623: the flags are only
    cleared in a branch a full path buffer never reaches).  With a small max_paths a read therefore sees the flags its
    predecessor on the channel left behind; the chunked path reproduces that (found by tests/dev/fuzz_parity.py, seed 5007: the
    fourth read of a channel did two get_neighbor calls more when the flags were cleared).  Reads of the fuzz case, one channel;
    the scenario is checked to be one where the carry-over matters (a fresh Mapper per read answers differently)."""
    from uncalled_amd.realtime import MapPoolOrd
    po = oracle_lib
    prefix, sim, n = carry_over_data(tmp_path)
    ix = capi.Index(prefix, lib=lib)
    oix = po.Index(prefix)
    p = capi.default_params(lib)
    p.max_paths, p.max_rep_copy, p.max_chunks, p.chunk_time = 60, 2, 2, 8000 / p.sample_rate
    pool = MapPoolOrd(ix, n_channels=1, params=p)
    om = po.Mapper(oix, to_oracle_params(p))
    om.set_max_chunks(2)
    off = sim["offsets"]
    cal = (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    want, alone = [], []
    for i in range(n):
        raw = sim["signal"][int(off[i]):int(off[i + 1])]
        pool.add_read(0, i, raw, cal, key=i)
        want.append(om.chunk_read(po.calibrate(raw, *cal), 8000)[0])
        fresh = po.Mapper(oix, to_oracle_params(p))
        fresh.set_max_chunks(2)
        alone.append(fresh.chunk_read(po.calibrate(raw, *cal), 8000)[0])
    got = {}
    while pool.running():
        for key, r in pool.update():
            got[key] = r["hit"]
    names_dev = ix.seq_names()
    for i in range(n):
        assert capi.hit_paf_cols(got[i], names_dev) == po.hit_paf_cols(want[i], oix.ref_names()), i
        for f in ("event_i", "n_nbr", "n_sa", "n_lf", "notes"):
            assert int(got[i][f]) == int(want[i][f]), (i, f)
    assert any(int(want[i]["n_nbr"]) != int(alone[i]["n_nbr"]) or int(want[i]["event_i"]) != int(alone[i]["event_i"]) for i in range(1, n))


def case_chunked_realtime_path(lib, oracle_lib, example, goldens, n_channels=3, n_reads=9, max_chunks=None, long_read=False):
    """Config 5's path: reads replayed chunk by chunk over a few channels (MapPoolOrd semantics) through
    unc_rt_process_chunks; per-channel state persists across chunks AND reads.  Checked against the oracle fed the
    same reads in the same per-channel order, and (default max_chunks, channel 0 order) against the reference goldens."""
    from uncalled_amd.realtime import MapPoolOrd
    po = oracle_lib
    dev_index = _index(lib, example)
    oix = po.Index(example["prefix"])
    p = capi.default_params(lib)
    if max_chunks:
        p.max_chunks = max_chunks
    pool = MapPoolOrd(dev_index, n_channels=n_channels, params=p)
    off = goldens["sim_offsets"]
    reads = [(example["signal"], (example["range"], example["offset"], example["digitisation"]))]
    for i in range(n_reads - 1):
        reads.append((goldens["sim_signal"][int(off[i]):int(off[i + 1])], (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)))
    if long_read:
        # an off-target read with more events than the 6000-slot normaliser ring holds: the ring wraps inside one read
        from tools.simulate_reads import simulate_reads
        sim = simulate_reads(np.zeros(20000, np.uint8), [20000], 1, seed=9, read_bases=5200, off_target=1.0)
        reads.insert(1, (sim["signal"], (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)))
    oms = [po.Mapper(oix) for _ in range(n_channels)]
    for om in oms:
        if max_chunks:
            om.set_max_chunks(max_chunks)
    want, fate = {}, {}
    for i, (raw, cal) in enumerate(reads):
        ch = i % n_channels
        pool.add_read(ch, i, raw, cal, key=i)
        want[i], used = oms[ch].chunk_read(po.calibrate(raw, *cal), 4000)
        fate[i] = (used, oms[ch].rt_ended())
    got = {}
    rounds = 0
    while pool.running():
        for key, r in pool.update():
            got[key] = r
        rounds += 1
        assert rounds < 1000
    names = dev_index.seq_names()
    for i in range(len(reads)):
        h, o = got[i]["hit"], want[i]
        assert int(h["status"]) == 0
        assert capi.hit_paf_cols(h, names) == po.hit_paf_cols(o, oix.ref_names()), i
        for f in ("event_i", "n_nbr", "n_sa", "n_lf", "notes"):
            assert int(h[f]) == int(o[f]), (i, f)
        assert got[i]["state"] == (capi.RT_MAPPED if o["mapped"] else capi.RT_FAILED), i
        assert (pool.chunks_used[i], bool(got[i]["ended"])) == fate[i], i        # chunks the read was given, Paf::ENDED
    if long_read:
        assert int(want[1]["event_i"]) > 6500 and not want[1]["mapped"]
    if n_channels == 1 and not max_chunks and not long_read:
        f = {str(n): j for j, n in enumerate(goldens["hit_fields"])}
        for i in range(len(reads)):
            for name in ("mapped", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "matches", "event_i", "n_nbr", "n_lf"):
                assert int(got[i]["hit"][name]) == int(goldens["chunk_hits"][i][f[name]]), (i, name)


def case_cluster_overflow_remap(lib, oracle_lib, example, goldens):
    """The reference's SeedTracker is unbounded; reads that outgrow the per-slot cluster array are re-mapped on the device
    with more room until they fit.  Forced here with an absurdly small array."""
    dev_index = _index(lib, example)
    oix = oracle_lib.Index(example["prefix"])
    n = 6
    off = goldens["sim_offsets"][:n + 1].copy()
    raw = goldens["sim_signal"][:int(off[n])]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    hits = capi.Mapper(dev_index, n_slots=2, max_clusters=8).map_batch(raw, off, cal)
    assert_hits_equal(hits, oracle_hits(oix, raw, off, cal), "remap")


def case_wide_sort_keys(lib, oracle_lib, example, goldens, monkeypatch, n=10):
    """References too large for the packed 64-bit sort key (GRCh38-sized: seq_len x k-mer range x 2^16 > 2^64) take
    the 128-bit key path; forced here through UNC_WIDE_KEYS on the small index."""
    monkeypatch.setenv("UNC_WIDE_KEYS", "1")
    ix = capi.Index(example["prefix"], lib=lib)
    monkeypatch.delenv("UNC_WIDE_KEYS", raising=False)
    oix = oracle_lib.Index(example["prefix"])
    off = goldens["sim_offsets"][:n + 1].copy()
    raw = goldens["sim_signal"][:int(off[n])]
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    hits = capi.Mapper(ix, n_slots=3).map_batch(raw, off, cal)
    assert_hits_equal(hits, oracle_hits(oix, raw, off, cal), "wide keys")


def case_sliced_scheduler(lib, oracle_lib, example, goldens, max_paths=10000, slice_events=37, n_slots=5, n_waves=2, n_reads=24):
    """More reads in flight than wavefronts (DevSched): reads are parked after `slice_events` events and resumed later,
    possibly by another wavefront; answers and work counters must not change (also with the max_paths cut-off, whose
    stale sources_added_ flags travel with the parked read)."""
    dev_index = _index(lib, example)
    p = capi.default_params(dev_index.L)
    p.max_paths = max_paths
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    m = capi.Mapper(dev_index, params=p, n_slots=n_slots, n_waves=n_waves, slice_events=slice_events)
    hits = m.map_batch(raw, off, cal)
    again = m.map_batch(raw, off, cal)          # the rings are re-initialised per batch
    for name in capi.RESULT_FIELDS:
        assert np.array_equal(again[name], hits[name]), name
    oix = oracle_lib.Index(example["prefix"])
    assert_hits_equal(hits, oracle_hits(oix, raw, off, cal, to_oracle_params(p), fresh_mapper_per_read=True), "sliced")


def case_scheduler_rings_per_xcd(lib, oracle_lib, example, goldens, n_reads=24, slice_events=37):
    """One pair of scheduler rings per XCD (SchedCtl): a slot stays with the wavefronts of one XCD, which park and resume its reads without
    an L2 write-back / invalidate.  512 slots divide into eight shares of 64: the library settles on the device's XCD count (8 on the
    MI355X; the emulator deals its workgroups to eight XCDs from XCD 3 on, so the shares in use are not the first), a mapper told
    sched_parts=1 keeps one pair for the device, and both answer as the oracle does; a share of fewer than 64 slots falls back to one pair."""
    dev_index = _index(lib, example)
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    oix = oracle_lib.Index(example["prefix"])
    want = oracle_hits(oix, raw, off, cal, fresh_mapper_per_read=True)
    m = capi.Mapper(dev_index, n_slots=512, n_waves=8, slice_events=slice_events, pool_chunks=256)
    parts = m.sched_parts()
    assert parts in (1, 8), parts               # (1: a device, or a partition mode, that shows one XCD)
    hits = m.map_batch(raw, off, cal)
    again = m.map_batch(raw, off, cal)
    assert_hits_equal(hits, want, "rings per XCD")
    m1 = capi.Mapper(dev_index, n_slots=512, n_waves=8, slice_events=slice_events, pool_chunks=256, sched_parts=1)
    assert m1.sched_parts() == 1
    hits1 = m1.map_batch(raw, off, cal)
    for name in capi.RESULT_FIELDS:
        assert np.array_equal(again[name], hits[name]) and np.array_equal(hits1[name], hits[name]), name
    m2 = capi.Mapper(dev_index, n_slots=256, n_waves=8, slice_events=slice_events, pool_chunks=256)      # shares of 32 slots
    assert m2.sched_parts() == 1
    with pytest.raises(capi.UncalledHipError):
        capi.Mapper(dev_index, n_slots=512, n_waves=8, slice_events=slice_events, pool_chunks=256, sched_parts=3)
    return parts


def case_cluster_pool_pressure(lib, oracle_lib, example, goldens, pool_chunks=2, n_waves=2, n_reads=12):
    """The nodes of all seed-cluster sets come from one pool.  With fewer chunks than reads in flight some reads find it
    dry; with a tiny allowance of nodes some outgrow that: both are mapped again after the batch (with fewer reads sharing
    the pool; with a larger allowance) and every read still answers as the oracle does.  A roomy pool needs no second pass."""
    dev_index = _index(lib, example)
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    oix = oracle_lib.Index(example["prefix"])
    want = oracle_hits(oix, raw, off, cal, fresh_mapper_per_read=True)
    m = capi.Mapper(dev_index, n_slots=3 * n_waves, n_waves=n_waves, slice_events=60, pool_chunks=pool_chunks)
    assert m.geometry() == dict(n_waves=n_waves, n_slots=3 * n_waves, slice_events=60, pool_chunks=pool_chunks, max_clusters=1 << 20)
    hits = m.map_batch(raw, off, cal)
    assert m.last_remap()[0] > 0            # 3 * n_waves reads in flight, pool_chunks < that many chunks
    assert_hits_equal(hits, want, "pool pressure")
    m2 = capi.Mapper(dev_index, n_slots=3 * n_waves, n_waves=n_waves, slice_events=60, pool_chunks=64, max_clusters=16)
    hits2 = m2.map_batch(raw, off, cal)      # one leaf per read: the ones with more than 64 clusters outgrow the directory
    assert m2.last_remap()[0] > 0
    m3 = capi.Mapper(dev_index, n_slots=3 * n_waves, n_waves=n_waves, slice_events=60, pool_chunks=64)
    hits3 = m3.map_batch(raw, off, cal)
    assert m3.last_remap()[0] == 0
    # both causes in one batch: reads that find the pool dry run again with fewer fellows, reads past their own allowance
    # (max_clusters / 4 nodes) with a larger one -- and a read may meet one cause after the other
    m4 = capi.Mapper(dev_index, n_slots=3 * n_waves, n_waves=n_waves, slice_events=60, pool_chunks=pool_chunks, max_clusters=16)
    hits4 = m4.map_batch(raw, off, cal)
    assert m4.last_remap()[0] > 0
    for name in capi.RESULT_FIELDS:
        assert np.array_equal(hits[name], hits2[name]) and np.array_equal(hits[name], hits3[name]) and np.array_equal(hits[name], hits4[name]), name
    # the pool is sized by NEED (unc_mapper_pool_usage): a named size stays as it is and its high-water mark is reported ...
    u = m.pool_usage()
    assert u["chunks"] == pool_chunks and u["resizes"] == 0 and u["high_water_ever"] == pool_chunks          # it ran dry: every chunk was out
    u3 = m3.pool_usage()
    assert u3["chunks"] == 64 and 0 < u3["high_water_last_batch"] <= 3 * n_waves and u3["resizes"] == 0      # at most a chunk per read in flight here
    # ... the default starts from the rule of thumb and is then cut to four times the most chunks that were out at once when it holds more
    # than eight times that (never below one per slot / 16; four times, since round 5 saw one batch's peak move by tens of per cent
    # between launches) -- but only by a batch that FILLED the slots (round-5 advice: a warm-up call or a handful of reads must not cut a
    # pool sized for a full load of reads in flight).  32 slots, twelve reads: the rule of thumb gives 256 chunks and the batch leaves them
    m5 = capi.Mapper(dev_index, n_slots=32, n_waves=n_waves, slice_events=60)
    before = m5.geometry()["pool_chunks"]
    hits5 = m5.map_batch(raw, off, cal)
    u5 = m5.pool_usage()
    assert before == 256 and 0 < u5["high_water_ever"] <= n_reads and m5.last_remap()[0] == 0
    assert (u5["chunks"], u5["resizes"]) == ((256, 0) if n_reads < 32 else (max(32, 4 * u5["high_water_ever"]), 1)), u5
    # as many slots as reads (at most six): the batch fills the slots, and the pool (eight chunks per slot by the rule of thumb) is cut
    # by the rule above whenever the rule says so
    ns7 = min(6, n_reads)
    m7 = capi.Mapper(dev_index, n_slots=ns7, n_waves=min(n_waves, ns7), slice_events=60)
    before7 = m7.geometry()["pool_chunks"]
    hits7 = m7.map_batch(raw, off, cal)
    u7 = m7.pool_usage()
    need = max(16, 4 * u7["high_water_ever"])        # never below 16 chunks / one chunk per slot
    want7 = need if 2 * need < before7 else before7
    assert before7 == max(16, 8 * ns7) and 0 < u7["high_water_ever"] <= ns7, (before7, u7)
    assert u7["chunks"] == want7 and u7["resizes"] == (1 if want7 != before7 else 0), (before7, u7, want7)
    assert m7.last_remap()[0] == 0
    hits8 = m7.map_batch(raw, off, cal)              # the second batch finds the pool sized and leaves it
    assert m7.pool_usage()["resizes"] == u7["resizes"] and m7.pool_usage()["chunks"] == u7["chunks"] and m7.last_remap()[0] == 0
    for name in capi.RESULT_FIELDS:
        assert np.array_equal(hits[name], hits7[name]) and np.array_equal(hits[name], hits8[name]), name
    hits6 = m5.map_batch(raw, off, cal)
    assert m5.pool_usage()["resizes"] == u5["resizes"] and m5.pool_usage()["chunks"] == u5["chunks"] and m5.last_remap()[0] == 0
    for name in capi.RESULT_FIELDS:
        assert np.array_equal(hits[name], hits5[name]) and np.array_equal(hits[name], hits6[name]), name


def case_big_forests(lib, oracle_lib, example, goldens, tmp_path, monkeypatch, wide_too=True):
    """Path forests of more than 512 children per event on the small index (permissive thresholds): the sorts beyond one
    register block -- 512-key blocks + stages through memory -- in both key modes, against the oracle."""
    import shutil
    for suf in (".amb", ".ann", ".bwt", ".pac", ".sa"):
        shutil.copy(str(example["prefix"]) + suf, str(tmp_path / ("loose" + suf)))
    (tmp_path / "loose.uncl").write_text("default\t-10.07,-5.5,-5.0,-4.6,-4.3,-4.1\t0.3\t115.000\n")
    prefix = tmp_path / "loose"
    n = 3
    off_all = goldens["sim_offsets"]
    raw = np.concatenate([goldens["sim_signal"][int(off_all[i]):int(off_all[i]) + 5000] for i in range(n)])
    off = (np.arange(n + 1) * 5000).astype(np.uint64)
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    want = oracle_hits(oracle_lib.Index(prefix), raw, off, cal)
    for wide in ((False, True) if wide_too else (False,)):
        if wide:
            monkeypatch.setenv("UNC_WIDE_KEYS", "1")
        ix = capi.Index(prefix, lib=lib)
        hits = capi.Mapper(ix, n_slots=n).map_batch(raw, off, cal)
        assert_hits_equal(hits, want, "big forests, wide keys" if wide else "big forests")
        assert (hits["n_nbr"] / np.maximum(hits["event_i"], 1)).min() > 1000        # ~600 parents, well over 512 children per event
    monkeypatch.delenv("UNC_WIDE_KEYS", raising=False)


def case_mid_reference(lib, oracle_lib, tmp_path, n=3, genome=800000, cut=8000):
    """A reference large enough that few children sit on a k-mer's boundary row (on the 10 kb example index nearly every
    event has one and takes the sort through memory): events with many children then go through the merge that walks its
    output tile in LDS.  Built here (synthetic genome, permissive thresholds), against the oracle."""
    from uncalled_amd.build_index import build_from_codes, synthetic_genome
    from tools.simulate_reads import simulate_reads
    names, lens, codes = synthetic_genome(1, genome, seed=11)
    prefix = tmp_path / "mid"
    build_from_codes(prefix, names, [""], lens, codes)
    (tmp_path / "mid.uncl").write_text("default\t-10.07,-5.5,-5.0,-4.6,-4.3,-4.1\t0.3\t115.000\n")     # permissive: hundreds of children per event
    sim = simulate_reads(codes, lens, n, seed=5)
    off = sim["offsets"]
    raw = np.concatenate([sim["signal"][int(off[i]):int(off[i]) + cut] for i in range(n)])
    o = (np.arange(n + 1) * cut).astype(np.uint64)
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    hits = capi.Mapper(capi.Index(prefix, lib=lib), n_slots=n).map_batch(raw, o, cal)
    want = oracle_hits(oracle_lib.Index(prefix), raw, o, cal)
    assert_hits_equal(hits, want, "mid reference")
    assert (hits["n_nbr"] / np.maximum(hits["event_i"], 1)).max() > 300      # events well past the merge threshold


def case_chunked_mid_reference(lib, oracle_lib, tmp_path, n=2, genome=800000, cut=8000, chunk_len=4000, n_channels=2):
    """The chunked path where events have HUNDREDS of children (case_mid_reference's reference and thresholds): the team kernel's own
    sort -- runs repaired by four waves at once, merge tiles dealt to the waves, the leader walking the sorted tiles in the other
    waves' LDS -- only runs for events past the merge threshold, which the 10 kb example index hardly produces.  Chunk by chunk
    against the oracle's chunk path, per-channel state carried from read to read."""
    from uncalled_amd.build_index import build_from_codes, synthetic_genome
    from uncalled_amd.realtime import MapPoolOrd
    from tools.simulate_reads import simulate_reads
    po = oracle_lib
    names, lens, codes = synthetic_genome(1, genome, seed=11)
    prefix = tmp_path / "mid"
    build_from_codes(prefix, names, [""], lens, codes)
    (tmp_path / "mid.uncl").write_text("default\t-10.07,-5.5,-5.0,-4.6,-4.3,-4.1\t0.3\t115.000\n")     # permissive: hundreds of children per event
    sim = simulate_reads(codes, lens, n, seed=5)
    off = sim["offsets"]
    dev_index = capi.Index(prefix, lib=lib)
    oix = po.Index(prefix)
    p = capi.default_params(lib)
    p.chunk_time = chunk_len / p.sample_rate
    pool = MapPoolOrd(dev_index, n_channels=n_channels, params=p)
    oms = [po.Mapper(oix, to_oracle_params(p)) for _ in range(n_channels)]
    cal = (CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    want, fate = {}, {}
    for i in range(n):
        raw = sim["signal"][int(off[i]):int(off[i]) + cut]
        pool.add_read(i % n_channels, i, raw, cal, key=i)
        want[i], used = oms[i % n_channels].chunk_read(po.calibrate(raw, *cal), chunk_len)
        fate[i] = (used, oms[i % n_channels].rt_ended())
    got, rounds = {}, 0
    while pool.running():
        for key, r in pool.update():
            got[key] = r
        rounds += 1
        assert rounds < 1000
    names_dev = dev_index.seq_names()
    for i in range(n):
        h, o = got[i]["hit"], want[i]
        assert int(h["status"]) == 0
        assert capi.hit_paf_cols(h, names_dev) == po.hit_paf_cols(o, oix.ref_names()), i
        for f in ("event_i", "n_nbr", "n_sa", "n_lf", "notes"):
            assert int(h[f]) == int(o[f]), (i, f)
        assert (pool.chunks_used[i], bool(got[i]["ended"])) == fate[i], i
    assert max(int(want[i]["n_nbr"]) / max(int(want[i]["event_i"]), 1) for i in range(n)) > 300      # events well past the merge threshold


def case_same_row_two_kmers(lib, oracle_lib, tmp_path, n=3, seed=5):
    """Two children with the SAME one-row range and DIFFERENT k-mers: BwaIndex::get_base_range starts one row low
    (bwa_index.hpp:172-174), so neighbouring k-mer ranges share a boundary row; the reference walks such a pair in seed_prob order
    (mapper.cpp:543-563), which a narrow sort key cannot express.  The narrow-key walk notices the pair, puts sources_added_ back and
    the event is walked again on 128-bit keys (k_map.hip: WalkState::mixed, phase_S_wide_redo).  On a 3 x 60 kb reference with
    permissive thresholds about one event in a hundred is such an event (none on the bench references' scale cases, none on the 10 kb
    example): the emulator build counts them and the case insists that some were seen."""
    import ctypes
    from uncalled_amd.build_index import build_from_codes, synthetic_genome
    from tools.simulate_reads import simulate_reads
    names, lens, codes = synthetic_genome(3, 60000, seed=77)
    prefix = tmp_path / "fz"
    build_from_codes(prefix, names, [""] * 3, lens, codes)
    (tmp_path / "fz.uncl").write_text("default\t-10.07,-4.6,-4.0,-3.6,-3.3,-3.1\t0.3\t115.000\n")
    try:
        cnt = ctypes.c_ulonglong.in_dll(lib, "unc_sim_wide_redo_count")       # (the emulator build only)
    except ValueError:
        cnt = None
    before = cnt.value if cnt is not None else 0
    dix, oix = capi.Index(prefix, lib=lib), oracle_lib.Index(prefix)
    cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    for off_target in (0.0, 1.0):
        sim = simulate_reads(codes, lens, n, seed=seed, read_bases=600, off_target=off_target)
        hits = capi.Mapper(dix, n_slots=n).map_batch(sim["signal"], sim["offsets"], cal)
        assert_hits_equal(hits, oracle_hits(oix, sim["signal"], sim["offsets"], cal), "same row, two k-mers (off target %.0f)" % off_target)
    import os
    if cnt is not None and not os.environ.get("UNC_WIDE_KEYS"):        # (with 128-bit keys forced every event is walked on them at once)
        assert cnt.value - before >= 10, "hardly any event took the wide-key redo (%d): the path is untested" % (cnt.value - before)


def case_batch_in_two_halves(lib, oracle_lib, example, goldens, n_reads=6):
    """unc_map_batch_begin / unc_map_batch_end: a batch launched and collected in two calls, two mappers over one index with a batch in
    flight on each (the second begun before the first is collected, as bench.py and a two-mapper worker loop do), host and device
    buffers alike -> the hits of unc_map_batch, which are the oracle's.  A second _begin before the _end, and an _end without a
    _begin, are refused."""
    dev_index = _index(lib, example)
    off_all = goldens["sim_offsets"]
    raw = goldens["sim_signal"][:int(off_all[n_reads])]
    off = off_all[:n_reads + 1].copy()
    cal = capi.make_calib(n_reads, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    want = oracle_hits(oracle_lib.Index(example["prefix"]), raw, off, cal, fresh_mapper_per_read=True)
    a = capi.Mapper(dev_index, n_slots=4, n_waves=2, slice_events=50)
    b = capi.Mapper(dev_index, n_slots=3, n_waves=3)
    whole = a.map_batch(raw, off, cal)
    assert_hits_equal(whole, want, "unc_map_batch")
    half = n_reads // 2
    off2 = (off[half:] - off[half]).astype(np.uint64)
    raw2 = raw[int(off[half]):]
    for _round in range(2):                    # (twice: a mapper is as good as new after _end)
        a.begin_batch(raw, off, cal)
        b.begin_batch(raw2, off2, cal[half:])  # begun while a's batch is in flight
        with pytest.raises(capi.UncalledHipError):
            a.begin_batch(raw, off, cal)       # one batch per mapper at a time
        ha = a.end_batch()
        hb = b.end_batch()
        for name in capi.RESULT_FIELDS:
            assert np.array_equal(ha[name], whole[name]), name
            assert np.array_equal(hb[name], whole[name][half:]), name
    with pytest.raises(capi.UncalledHipError):
        a.end_batch()                          # nothing begun
    assert_hits_equal(a.map_batch(raw, off, cal), want, "unc_map_batch after the halves")
