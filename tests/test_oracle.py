"""CPU tests of the test oracle itself: the plain-C restatement (oracle/unc_oracle.c) is pinned against
(a) the committed goldens generated from the reference's own object code (tests/golden/make_goldens.py), and
(b) that object code live, wherever /root/reference exists (this container)."""
from pathlib import Path

import numpy as np
import pytest

from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE


def _gold_hit(goldens, row):
    return dict(zip([str(x) for x in goldens["hit_fields"]], [int(v) for v in row]))


def test_calibration_and_events_match_golden(oracle_lib, example, goldens):
    po = oracle_lib
    sig = po.calibrate(example["signal"], example["range"], example["offset"], example["digitisation"])
    assert np.array_equal(sig, goldens["ex_calibrated"])
    ev, mel, tot = po.detect_events(sig)
    g = goldens["ex_events"]
    assert len(ev) == len(g) == 6171
    for f in ("mean", "stdv", "start", "length"):
        assert np.array_equal(ev[f], g[f]), f
    assert np.float32(mel) == goldens["ex_mean_event_len"] and tot == int(goldens["ex_total_events"])


def test_model_normaliser_matchprob_match_golden(oracle_lib, goldens):
    po = oracle_lib
    a, b, c, mm, ms = po.model_tables()
    assert np.array_equal(a, goldens["model_means"]) and np.array_equal(b, goldens["model_vars_x2"])
    assert np.array_equal(c, goldens["model_lognorm"])
    assert np.float32(mm) == goldens["model_mean"] and np.float32(ms) == goldens["model_stdv"]
    lv, sc, sh = po.normalize(goldens["ex_events"]["mean"])
    assert np.array_equal(lv, goldens["ex_levels"])      # north star: within 1e-5; achieved: bit-exact
    assert np.float32(sc) == goldens["ex_scale"] and np.float32(sh) == goldens["ex_shift"]
    for i in range(64):
        assert np.array_equal(po.match_probs(lv[i]), goldens["ex_probs"][i])


def test_index_tables_match_golden(oracle_lib, example, goldens):
    ix = oracle_lib.Index(example["prefix"])
    assert np.array_equal(ix.kmer_ranges(), goldens["kmer_ranges"])
    assert np.array_equal(ix.thresholds(), goldens["thresholds"])
    # the .uncl line of the example index: bins 63..58 then copies (mapper.cpp:146-156)
    thr = ix.thresholds()
    assert thr[63] == np.float32(-10.07) and thr[57] == thr[0] == np.float32(-2.2677272727272726)


def test_fm_index_against_naive_suffix_array(oracle_lib, example):
    """minibwa pinned on the reference's prebuilt index: SA(k) for every row equals the naive suffix array
    of fwd+revcomp decoded from .pac; block counts reproduce L2."""
    ix = oracle_lib.Index(example["prefix"])
    pac = np.fromfile(str(example["prefix"]) + ".pac", dtype=np.uint8)
    n = 10000
    codes = np.array([(pac[i >> 2] >> ((~i & 3) << 1)) & 3 for i in range(n)], dtype=np.uint8)
    t = np.concatenate((codes, 3 - codes[::-1]))
    assert ix.size == 2 * n
    s = bytes(t + 1)   # 1..4 so that the empty suffix sorts first
    sa = sorted(range(2 * n + 1), key=lambda i: s[i:])
    got = [ix.sa(k) for k in range(1, 2 * n + 1)]
    assert got == sa[1:]
    # backward search of a known 12-mer lands on rows whose SA values are its occurrences
    q = t[5000:5012]
    lo, hi = 0, 2 * n   # whole matrix: rows of suffixes starting with q[-1] ...
    # build the range base by base the way BwaIndex does (bwa_index.hpp:124-132, 158-174)
    counts = np.bincount(t, minlength=4)
    L2 = np.concatenate(([0], np.cumsum(counts)))
    lo, hi = int(L2[q[-1]]), int(L2[q[-1] + 1])
    for b in q[-2::-1]:
        lo, hi = ix.get_neighbor(lo, hi, int(b))
    occ = sorted(ix.sa(k) for k in range(lo, hi + 1))
    want = [i for i in range(2 * n - 11) if np.array_equal(t[i:i + 12], q)]
    assert set(want) <= set(occ) and len(occ) - len(want) <= 1   # get_base_range starts one row low (SURVEY a-8)


def test_example_read_paf_matches_golden(oracle_lib, example, goldens):
    po = oracle_lib
    ix = po.Index(example["prefix"])
    h = po.Mapper(ix).map_read(goldens["ex_calibrated"])
    g = _gold_hit(goldens, goldens["ex_hit"])
    for f, v in g.items():
        assert int(h[f]) == v, f
    # the survey's probe of the current reference: 106 73 106 - ... 10000 6938 6976 38 39 255
    assert po.hit_paf_cols(h, ix.ref_names()) == (106, 73, 106, "-", "Escherichia_coli_chromosome:2400000-2410000",
                                                   10000, 6938, 6976, 38, 39, 255)


def test_synthetic_reads_match_golden(oracle_lib, example, goldens):
    po = oracle_lib
    ix = po.Index(example["prefix"])
    om = po.Mapper(ix)
    off = goldens["sim_offsets"]
    for i in range(len(off) - 1):
        sig = po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        h = om.map_read(sig)
        g = _gold_hit(goldens, goldens["sim_hits"][i])
        for f, v in g.items():
            assert int(h[f]) == v, (i, f)
        assert np.float32(h["mean_event_len"]) == goldens["sim_mel"][i]
    assert int(goldens["sim_hits"][:, 0].sum()) >= 40   # the simulator produces mappable reads


def test_simulated_reads_map_to_their_origin(goldens):
    """Sanity of the data tooling: mapped reads land on the simulated strand and locus."""
    hits = goldens["sim_hits"]
    f = {str(n): j for j, n in enumerate(goldens["hit_fields"])}
    ok = 0
    for i, row in enumerate(hits):
        if not row[f["mapped"]] or goldens["sim_contig"][i] < 0:
            continue
        pos, strand = int(goldens["sim_pos"][i]), int(goldens["sim_strand"][i])
        assert bool(row[f["fwd"]]) == (strand == 0)
        assert pos - 50 <= row[f["rf_st"]] <= pos + 1500 + 50
        ok += 1
    assert ok >= 35


def test_parameter_variants_match_golden(oracle_lib, example, goldens):
    """The oracle on the parameter sets of PARAM_VARIANTS against tests/golden/param_variant_goldens.npz (the reference's
    outputs, generated by tests/golden/make_param_goldens.py): runs where the reference is absent."""
    from tests.parity_cases import PARAM_VARIANTS, variant_params
    po = oracle_lib
    g = np.load(Path(__file__).resolve().parent / "golden" / "param_variant_goldens.npz")
    assert list(g["variants"]) == [repr(sorted(v.items())) for v in PARAM_VARIANTS], "regenerate the goldens: the variant list changed"
    ix = po.Index(example["prefix"])
    off = goldens["sim_offsets"]
    sigs = [goldens["ex_calibrated"]] + [
        po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        for i in range(8)]
    for vi, ov in enumerate(PARAM_VARIANTS):
        om = po.Mapper(ix, variant_params(po.default_params(), ov))
        for i, sig in enumerate(sigs):
            h = om.map_read(sig)
            for f, v in _gold_hit(goldens, g["hits"][vi, i]).items():
                assert int(h[f]) == v, (ov, i, f)


def test_parameter_variants_equal_live_reference(oracle_lib, ref_lib, example, goldens):
    """The oracle against the reference's object code on parameter sets away from the defaults (tests/parity_cases.py
    PARAM_VARIANTS: repeat-seed limits, stay limits, seed probability, max_events, the tracker's confidence thresholds, the
    detector's thresholds, a small max_paths with long stays) -- the sets the device is then compared with the oracle on."""
    from tests.parity_cases import PARAM_VARIANTS, variant_params
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    ix = po.Index(example["prefix"])
    off = goldens["sim_offsets"]
    sigs = [goldens["ex_calibrated"]] + [
        po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        for i in range(8)]
    try:
        n_unmapped = 0
        for ov in PARAM_VARIANTS:
            p = variant_params(po.default_params(), ov)
            pr.set_params(p)
            om, rm = po.Mapper(ix, p), pr.Mapper()
            for i, sig in enumerate(sigs):
                h, r = om.map_read(sig), rm.map_read(sig)
                assert po.hit_paf_cols(h, ix.ref_names()) == r.paf_cols(), (ov, i)
                assert (int(h["event_i"]), int(h["n_events"]), int(h["n_nbr"]), int(h["n_sa"]), int(h["n_lf"])) == \
                       (r.event_i, r.n_events, r.n_nbr, r.n_sa, r.n_lf), (ov, i)
                n_unmapped += 0 if int(h["mapped"]) else 1
        assert n_unmapped > 0          # (some variant must have changed an outcome, or the sets test nothing)
    finally:
        pr.set_params(po.default_params())


@pytest.mark.parametrize("max_paths", [10000, 300, 130, 97])
def test_oracle_equals_live_reference(oracle_lib, ref_lib, example, goldens, max_paths):
    """Live comparison with the reference's object code, incl. the max_paths cut-off logic (mapper.cpp:480,507,521)
    exercised with a small buffer.  ONE Mapper on either side maps the reads back to back: with 130 and 97 paths sources_added_
    flags left by one read change the next one's answer (tests/parity_cases.py:case_read_order_t1 relies on exactly that), so the
    carry-over of `uncalled map -t 1` is pinned here on the reference's own Mapper.  Skipped where /root/reference is absent."""
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    pr.set_max_paths(max_paths)
    p = po.default_params()
    p.max_paths = max_paths
    ix = po.Index(example["prefix"])
    om, rm = po.Mapper(ix, p), pr.Mapper()
    off = goldens["sim_offsets"]
    sigs = [goldens["ex_calibrated"]] + [
        po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        for i in range(16)]
    for i, sig in enumerate(sigs):
        h, r = om.map_read(sig), rm.map_read(sig)
        assert po.hit_paf_cols(h, ix.ref_names()) == r.paf_cols(), i
        assert (int(h["event_i"]), int(h["n_nbr"]), int(h["n_sa"]), int(h["n_lf"])) == (r.event_i, r.n_nbr, r.n_sa, r.n_lf), i
    pr.set_max_paths(10000)


def test_oracle_trace_equals_live_reference(oracle_lib, ref_lib, example, goldens):
    """Per-event path buffers and seed clusters agree with the reference after every map_next."""
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    ix = po.Index(example["prefix"])
    om, rm = po.Mapper(ix), pr.Mapper()
    sig = goldens["ex_calibrated"]
    steps = 0
    for (od, oe, opaths, oclus, omm, ols, onl), (rd, re_, rpaths, rclus, rmm, rls, rnl) in zip(om.trace(sig), rm.trace(sig)):
        assert od == rd and oe == re_
        assert len(opaths) == len(rpaths)
        for f in ("fm_start", "fm_end", "kmer", "length"):
            assert np.array_equal(opaths[f], rpaths[f]), (steps, f)
        v = opaths["length"] > 0
        for f in ("event_moves", "seed_prob", "consec_stays", "sa_checked"):
            assert np.array_equal(opaths[f][v], rpaths[f][v]), (steps, f)
        for j in np.flatnonzero(v)[:50]:
            L = int(opaths["length"][j])
            assert np.array_equal(opaths["prob_sums"][j][:L + 1], rpaths["prob_sums"][j][:L + 1])
        assert np.array_equal(oclus, rclus), steps
        assert ols == rls and onl == rnl
        if omm["total_len"] or rmm["total_len"]:
            assert omm == rmm
        steps += 1
    assert steps == 178


CHUNK_FIELDS = ("mapped", "fwd", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "rf_len", "matches", "event_i", "n_nbr", "n_sa", "n_lf")


def _chunk_signals(oracle_lib, goldens):
    po = oracle_lib
    off = goldens["sim_offsets"]
    return [goldens["ex_calibrated"]] + [
        po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION) for i in range(30)]


def test_chunked_path_matches_golden(oracle_lib, example, goldens):
    """Mapper's chunk API (rolling normaliser, EventProfiler, 5-event batches) on one channel, 31 reads in a row."""
    po = oracle_lib
    ix = po.Index(example["prefix"])
    om = po.Mapper(ix)
    f = {str(n): j for j, n in enumerate(goldens["hit_fields"])}
    for i, sig in enumerate(_chunk_signals(po, goldens)):
        h, used = om.chunk_read(sig, 4000)
        for name in CHUNK_FIELDS:
            assert int(h[name]) == int(goldens["chunk_hits"][i][f[name]]), (i, name)
        assert used == int(goldens["chunk_used"][i]), i
    # the survey's probe of the chunk path on the example read: 67 41 67 - ... 6948 6977 29 30, 107 events
    g0 = goldens["chunk_hits"][0]
    assert (g0[f["rd_len"]], g0[f["rd_st"]], g0[f["rd_en"]], g0[f["rf_st"]], g0[f["rf_en"]], g0[f["matches"]], g0[f["event_i"]]) == \
        (67, 41, 67, 6948, 6977, 29, 107)


@pytest.mark.parametrize("max_chunks", [1000000, 2, 1])
def test_chunked_path_equals_live_reference(oracle_lib, ref_lib, example, goldens, max_chunks):
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    pr.lib().ref_set_max_chunks(max_chunks)
    ix = po.Index(example["prefix"])
    om, rm = po.Mapper(ix), pr.Mapper()
    om.set_max_chunks(max_chunks)
    for i, sig in enumerate(_chunk_signals(po, goldens)[:12]):
        (h, hu), (r, ru) = om.chunk_read(sig, 4000), rm.chunk_read(sig, 4000, i)
        assert po.hit_paf_cols(h, ix.ref_names()) == r.paf_cols(), i
        assert (hu, int(h["event_i"]), int(h["n_nbr"]), int(h["n_sa"]), int(h["n_lf"])) == (ru, r.event_i, r.n_nbr, r.n_sa, r.n_lf), i
        assert om.rt_ended() == bool(pr.lib().ref_last_ended()), i          # Paf::ENDED (mapper.cpp:386)
    pr.lib().ref_set_max_chunks(1000000)


def test_chunked_stage_tap_equals_live_reference(oracle_lib, ref_lib, example, goldens):
    """Below the PAF: Mapper::evdt_ / evt_prof_ / norm_ of the reference after every read of a channel (EventDetector counters,
    the EventProfiler's window and queue, the rolling Normalizer's mean / varsum / ring) against the oracle's, bit for bit."""
    from tests.parity_cases import assert_rt_taps_equal
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    ix = po.Index(example["prefix"])
    om, rm = po.Mapper(ix), pr.Mapper()
    for i, sig in enumerate(_chunk_signals(po, goldens)[:12]):
        om.chunk_read(sig, 4000)
        rm.chunk_read(sig, 4000, i)
        (a, ra), (b, rb) = om.rt_tap(), rm.rt_tap()
        assert_rt_taps_equal(a, ra, b, rb, "read %d" % i)
    assert int(a["norm_n"]) == 6000


def test_chunked_flags_carry_over_equals_live_reference(oracle_lib, tmp_path):
    """sources_added_ survives Mapper::new_read: with a small max_paths the reads of one channel influence each other.  The
    oracle against the reference on the scenario where that changes the work counters (tests/parity_cases.py carry_over_data).
    In a process of its own: the reference keeps its index in statics, and this case needs another one than the other tests."""
    import subprocess
    import sys
    from oracle import pyref
    if not pyref.available():
        pytest.skip("oracle/_ref not built")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from pathlib import Path\n"
        "from oracle import pyoracle as po, pyref as pr\n"
        "from tests.parity_cases import carry_over_data, CARRY_OVER_PARAMS, variant_params\n"
        "from tools.simulate_reads import CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION\n"
        "prefix, sim, n = carry_over_data(Path(%r))\n"
        "p = variant_params(po.default_params(), {k: v for k, v in CARRY_OVER_PARAMS.items() if k != 'max_chunks'})\n"
        "pr.init(prefix); pr.set_params(p); pr.lib().ref_set_max_chunks(2)\n"
        "ix = po.Index(prefix); om = po.Mapper(ix, p); om.set_max_chunks(2); rm = pr.Mapper()\n"
        "off = sim['offsets']\n"
        "for i in range(n):\n"
        "    sig = po.calibrate(sim['signal'][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)\n"
        "    (h, hu), (r, ru) = om.chunk_read(sig, 8000), rm.chunk_read(sig, 8000, i)\n"
        "    assert po.hit_paf_cols(h, ix.ref_names()) == r.paf_cols(), i\n"
        "    assert (hu, int(h['event_i']), int(h['n_nbr']), int(h['n_sa']), int(h['n_lf'])) == (ru, r.event_i, r.n_nbr, r.n_sa, r.n_lf), i\n"
        "print('CARRY_OVER_OK')\n") % (str(Path(__file__).resolve().parents[1]), str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "CARRY_OVER_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_chunked_path_variants_equal_live_reference(oracle_lib, ref_lib, example, goldens):
    """The chunked path of the oracle against the reference's (Mapper::new_read(Chunk&) / add_chunk / process_chunk / map_chunk)
    on the parameter sets and chunk lengths of tests/parity_cases.py CHUNK_VARIANTS, reads one after the other on ONE Mapper."""
    from tests.parity_cases import CHUNK_VARIANTS, variant_params
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    ix = po.Index(example["prefix"])
    try:
        for ov, chunk_len in CHUNK_VARIANTS:
            p = variant_params(po.default_params(), ov)
            pr.set_params(p)
            om, rm = po.Mapper(ix, p), pr.Mapper()
            for i, sig in enumerate(_chunk_signals(po, goldens)[:8]):
                (h, hu), (r, ru) = om.chunk_read(sig, chunk_len), rm.chunk_read(sig, chunk_len, i)
                assert po.hit_paf_cols(h, ix.ref_names()) == r.paf_cols(), (ov, chunk_len, i)
                assert (hu, int(h["event_i"]), int(h["n_nbr"]), int(h["n_sa"]), int(h["n_lf"])) == (ru, r.event_i, r.n_nbr, r.n_sa, r.n_lf), (ov, chunk_len, i)
    finally:
        pr.set_params(po.default_params())


def test_edge_case_reads_equal_live_reference(oracle_lib, ref_lib, example):
    """The reads of parity_cases.case_events_edge_cases (shorter than the detector windows, flat, negative samples, every
    head / tail length of the device kernel's blocked loop) through the reference's own EventDetector / Normalizer and,
    where a read yields events at all, through Mapper::map_read: the oracle the kernels are compared with on these reads is
    itself pinned on them.  Skipped where /root/reference is absent."""
    po, pr = oracle_lib, ref_lib
    pr.init(example["prefix"])
    rng = np.random.default_rng(5)
    reads = [np.array([500], np.int16), rng.integers(300, 700, 12).astype(np.int16),
             np.full(400, 512, np.int16), rng.integers(-200, 900, 700).astype(np.int16),
             np.repeat(rng.integers(350, 650, 60), 9).astype(np.int16)]
    for ln in list(range(13, 42)) + [63, 64, 65, 127, 129, 1000, 1003]:
        reads.append(np.repeat(rng.integers(330, 680, ln // 5 + 1), 5)[:ln].astype(np.int16) + rng.integers(-6, 7, ln).astype(np.int16))
    ix = po.Index(example["prefix"])
    om, rm = po.Mapper(ix), pr.Mapper()
    mapped_some = 0
    for i, r in enumerate(reads):
        sig = po.calibrate(r, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        ev, mel, tot = po.detect_events(sig)
        rev, rmel, rtot = pr.events(sig)
        assert len(ev) == len(rev) and tot == rtot, i
        assert np.array_equal(ev["mean"], rev["mean"]), i
        if len(ev) == 0:
            continue
        assert np.float32(mel) == np.float32(rmel), i
        lv, sc, sh = po.normalize(ev["mean"])
        rlv, _, _ = pr.norm_levels(ev["mean"])         # levels = Normalizer::pop -> at() (normalizer.cpp:114-118), what map_next consumes;
        assert np.array_equal(lv, rlv, equal_nan=True), i   # get_scale() divides by a float-rounded stdv and is not on the mapping path
        if not np.all(np.isfinite(lv)):
            continue                                   # a flat or one-event read: no levels to map
        h, q = om.map_read(sig), rm.map_read(sig)
        assert po.hit_paf_cols(h, ix.ref_names()) == q.paf_cols(), i
        assert (int(h["event_i"]), int(h["n_nbr"]), int(h["n_sa"]), int(h["n_lf"])) == (q.event_i, q.n_nbr, q.n_sa, q.n_lf), i
        mapped_some += 1
    assert mapped_some > 20


def test_tie_order_of_the_unstable_sort_is_counted_not_assumed(ref_lib, example, goldens):
    """mapper.cpp:531 sorts the children with orlp/pdqsort (unstable, un-vendored).  oracle/shim/pdqsort.h offers three tie orders at
    run time -- creation order (the project's convention), pattern-defeating quicksort restated from its published algorithm, and
    creation order reversed -- and counts the ties.  Pinned here: the restated sort sorts (duplicates, sorted / reversed /
    organ-pipe inputs, sizes around its thresholds 24 and 128); ties DO occur on the golden reads (one event in ten has one); and no
    PAF line of the 48 reads depends on their order.  (Work counters of a few reads do: a tie can decide which of two lineages with
    the same range and seed probability lives on.  bench.py reports the same counts on its own reads.)"""
    pr = ref_lib
    bad = [(n, s, k, sh) for n in (0, 1, 2, 23, 24, 25, 127, 128, 129, 130, 1000, 20000) for s in range(3)
           for k in (1, 2, 7, 1000, 1 << 30) for sh in range(4) if pr.sort_selftest(n, s, k, sh)]
    assert not bad, bad
    pr.init(example["prefix"])
    from oracle import pyoracle as po
    off = goldens["sim_offsets"]
    n = off.size - 1
    sig = po.calibrate(goldens["sim_signal"][:int(off[n])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    seen = {}
    try:
        for mode in (pr.SORT_STABLE, pr.SORT_PDQ_RESTATED, pr.SORT_REVERSED_TIES):
            pr.set_sort_mode(mode)
            pr.sort_stats(reset=True)
            hits, _ = pr.map_batch(sig, off[:n + 1], 4)
            seen[mode] = ([h.paf_cols() for h in hits], [(h.event_i, h.n_nbr, h.n_sa) for h in hits], pr.sort_stats())
    finally:
        pr.set_sort_mode(pr.SORT_STABLE)
    sorts, tie_events, tie_pairs = seen[pr.SORT_STABLE][2]
    assert sorts > 10000 and tie_events > sorts // 20 and tie_pairs >= tie_events
    for mode in (pr.SORT_PDQ_RESTATED, pr.SORT_REVERSED_TIES):
        assert seen[mode][0] == seen[pr.SORT_STABLE][0], mode
        assert seen[mode][2][0] == sorts
    # the order is not without effect: some read's work counters move under the reversed order
    assert seen[pr.SORT_REVERSED_TIES][1] != seen[pr.SORT_STABLE][1]
