"""The realtime host classes (Chunk / RealtimePool / ClientSim of `_uncalled_amd`, mirroring src/chunk.hpp:47-59,
src/realtime_pool.hpp:63-70, src/client_sim.hpp:39-44) -- their sources linked against the lanesim build of the C ABI and
driven the way MapPoolOrd drives the reference's pool (map_pool_ord.cpp:61-112: try_add_chunk, an empty chunk once a read
has run out, update), checked read by read against the oracle's chunked path."""
from pathlib import Path

import numpy as np
import pytest

from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE

G = Path(__file__).resolve().parent / "golden"


def case_chunk_class(unc):
    sig = [float(i) for i in range(10)]
    c = unc.Chunk("r1", 7, 3, 4000, sig, 2, 5)
    assert (c.id, c.channel, c.number, c.size(), c.empty(), c.start) == ("r1", 7, 3, 5, False, 4000)
    assert c.pop() == [2.0, 3.0, 4.0, 5.0, 6.0] and c.empty()
    assert unc.Chunk("r1", 7, 3, 0, sig, 8, 5).size() == 2                 # cut at the end of the read (chunk.cpp:60-66)
    raw = np.array([-3, 0, 700, 32767], dtype=np.int16)
    assert unc.Chunk("r", 1, 0, 0, "int16", raw.tobytes()).pop() == [-3.0, 0.0, 700.0, 32767.0]   # no calibration (chunk.cpp:33-38)
    f = np.array([1.5, -2.25], dtype=np.float32)
    assert unc.Chunk("r", 1, 0, 0, "float32", f.tobytes()).pop() == [1.5, -2.25]
    a, b = unc.Chunk("a", 1, 1, 0, sig, 0, 3), unc.Chunk("b", 2, 2, 9, sig, 0, 4)
    a.swap(b)
    assert (a.id, a.channel, a.number, a.size(), b.id, b.size()) == ("b", 2, 2, 4, "a", 3)


def _conf(unc, n_channels, **kw):
    c = unc.Conf()
    c.bwa_prefix = str(G / "example_index" / "example_ref")
    c.num_channels = n_channels
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def case_realtime_pool_ordered_replay(unc, po, example, goldens, n_reads=9):
    n_channels, chunk_len = 3, 4000
    off = goldens["sim_offsets"]
    reads = [po.calibrate(example["signal"], example["range"], example["offset"], example["digitisation"])]
    for i in range(n_reads - 1):
        reads.append(po.calibrate(goldens["sim_signal"][int(off[i]):int(off[i + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
    oix = po.Index(G / "example_index" / "example_ref")
    oms = [po.Mapper(oix) for _ in range(n_channels)]
    want = {i: oms[i % n_channels].chunk_read(reads[i], chunk_len)[0] for i in range(n_reads)}

    pool = unc.RealtimePool(_conf(unc, n_channels))
    queues = [[i for i in range(n_reads) if i % n_channels == ch] for ch in range(n_channels)]
    chunk_i = [0] * n_channels
    got = {}
    rounds = 0
    while any(queues) or not pool.all_finished():
        for ch, nm, paf in pool.update():
            i = nm
            got[i] = paf
            assert queues[ch - 1] and queues[ch - 1][0] == i
            queues[ch - 1].pop(0)
            chunk_i[ch - 1] = 0
        for ch in range(n_channels):                       # MapPoolOrd::update: the next chunk of the channel's front read
            if not queues[ch]:
                continue
            i = queues[ch][0]
            sig = reads[i]
            st = min(chunk_i[ch] * chunk_len, len(sig))
            c = unc.Chunk("read%d" % i, ch + 1, i, st, sig.tolist(), st, chunk_len)
            if pool.try_add_chunk(c):
                chunk_i[ch] += 1
        rounds += 1
        assert rounds < 2000
    names = oix.ref_names()
    assert sorted(got) == list(range(n_reads))
    for i in range(n_reads):
        cols = str(got[i]).split("\t")
        o = want[i]
        assert cols[0] == "read%d" % i
        wantcols = po.hit_paf_cols(o, names)
        if o["mapped"]:
            assert got[i].is_mapped()
            assert (int(cols[1]), int(cols[2]), int(cols[3]), cols[4], cols[5], int(cols[6]), int(cols[7]), int(cols[8]), int(cols[9]),
                    int(cols[10]), int(cols[11])) == wantcols, (i, cols, wantcols)
        else:
            assert not got[i].is_mapped() and int(cols[1]) == wantcols[0] and cols[2] == "*"
        assert "ch:i:%d" % (i % n_channels + 1) in cols
    pool.stop_all()
    assert pool.all_finished() and pool.update() == []


def case_add_chunk_resets_the_previous_read(unc, po, goldens):
    """RealtimePool::add_chunk (realtime_pool.cpp:74-110): a chunk of a NEW read on a channel whose read is still undecided
    ends that read (unmapped, `ended`) and starts the new one."""
    pool = unc.RealtimePool(_conf(unc, 2))
    noise = (np.random.default_rng(3).normal(90.0, 12.0, 9000)).astype(np.float32).tolist()      # never maps
    assert pool.add_chunk(unc.Chunk("noise", 1, 5, 100, noise, 0, 4000))
    assert not pool.add_chunk(unc.Chunk("noise", 1, 5, 4100, noise, 4000, 4000))                 # previous chunk not mapped yet
    assert pool.update() == [] and pool.active_count() == 1
    assert pool.add_chunk(unc.Chunk("noise", 1, 5, 4100, noise, 4000, 4000))
    assert pool.update() == []
    off = goldens["sim_offsets"]
    good = po.calibrate(goldens["sim_signal"][int(off[0]):int(off[1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION).tolist()
    assert pool.add_chunk(unc.Chunk("good", 1, 6, 0, good, 0, 4000))                              # new read number on channel 1
    out = pool.update()
    ended = [(ch, nm, p) for ch, nm, p in out if nm == 5]
    assert len(ended) == 1 and ended[0][0] == 1 and ended[0][2].is_ended() and not ended[0][2].is_mapped()
    assert str(ended[0][2]).split("\t")[0] == "noise" and int(str(ended[0][2]).split("\t")[1]) == int(8000 * 450.0 / 4000.0)
    k = 1
    mapped = [p for ch, nm, p in out if nm == 6]
    while not mapped and k * 4000 < len(good):
        assert pool.add_chunk(unc.Chunk("good", 1, 6, k * 4000, good, k * 4000, 4000))
        mapped = [p for ch, nm, p in pool.update() if nm == 6]
        k += 1
    assert mapped and mapped[0].is_mapped() and pool.all_finished()


def case_oversized_chunk_is_refused(unc):
    """A chunk longer than chunk_time * sample_rate (a client on another chunk time) is refused by add_chunk / try_add_chunk
    and leaves the pool running: the reference's ReadBuffer would append it, the device side stages one chunk_len per
    channel, and one bad chunk must not end the run."""
    pool = unc.RealtimePool(_conf(unc, 2))
    noise = (np.random.default_rng(4).normal(90.0, 12.0, 12000)).astype(np.float32).tolist()
    assert not pool.add_chunk(unc.Chunk("big", 1, 1, 0, noise, 0, 4001))
    assert not pool.try_add_chunk(unc.Chunk("big", 1, 1, 0, noise, 0, 8000))
    assert pool.update() == [] and pool.active_count() == 0 and pool.all_finished()
    assert pool.refused_chunks() == 2
    assert pool.add_chunk(unc.Chunk("ok", 1, 2, 0, noise, 0, 4000))           # the channel still takes a regular chunk
    assert pool.update() == [] and pool.active_count() == 1
    # ... and an oversized one mid-read is refused AND ends the read (unmapped + ended at the next update): the channel is free again
    assert not pool.add_chunk(unc.Chunk("ok", 1, 2, 4000, noise, 4000, 5000))
    assert pool.active_count() == 0 and pool.refused_chunks() == 3
    out = pool.update()
    assert len(out) == 1 and out[0][0] == 1 and out[0][1] == 2 and out[0][2].is_ended() and not out[0][2].is_mapped()
    assert str(out[0][2]).split("\t")[0] == "ok" and int(str(out[0][2]).split("\t")[1]) == int(4000 * 450.0 / 4000.0)
    assert pool.all_finished()
    assert pool.add_chunk(unc.Chunk("next", 1, 3, 0, noise, 0, 4000)) and pool.active_count() == 1
    # two reads given up on ONE channel between two updates (round-4 advice): read 3 is mapping; a chunk of read 4 takes the channel
    # over (3 is reset); an oversized chunk of read 4 arrives before update() and ends 4 as well.  Both get their unmapped + ended line
    # -- the reference reports every reset read (a single give-up record per channel lost read 3's)
    assert pool.update() == [] and pool.active_count() == 1
    assert pool.add_chunk(unc.Chunk("four", 1, 4, 0, noise, 0, 4000))
    assert not pool.add_chunk(unc.Chunk("four", 1, 4, 4000, noise, 4000, 5000))
    out = pool.update()
    assert sorted((ch, nm, str(p).split("\t")[0], p.is_ended(), p.is_mapped()) for ch, nm, p in out) == \
        [(1, 3, "next", True, False), (1, 4, "four", True, False)]
    assert pool.all_finished() and pool.update() == []


def case_client_sim_feeds_the_decision_loop(unc, tmp_path, goldens):
    """ClientSim-shaped source over fast5 files + the enrich/deplete loop of scripts/uncalled:216-256 (uncalled_amd sim)."""
    off = goldens["sim_offsets"]
    reads = [dict(id="sim-%d" % i, channel=1 + i % 2, number=i, start=1000 * i, range=CAL_RANGE, offset=CAL_OFFSET, digitisation=CAL_DIGITISATION,
                  signal=goldens["sim_signal"][int(off[i]):int(off[i + 1])].tolist()) for i in range(4)]
    f5 = tmp_path / "reads.fast5"
    assert unc.write_fast5(str(f5), reads, True, 4000.0)
    conf = _conf(unc, 2)
    client = unc.ClientSim(conf)
    client.add_fast5(str(f5))
    client.load_fast5s()
    assert client.run() and client.is_running
    first = client.get_read_chunks()
    assert [(ch, c.id, c.number, c.size()) for ch, c in first] == [(1, "sim-0", 0, 4000), (2, "sim-1", 1, 4000)]
    assert first[0][1].start == 0 and first[1][1].start == 1000
    client.stop_receiving_read(1, 0)                       # channel 1 moves on to its next read
    second = client.get_read_chunks()
    assert [(ch, c.id, c.start) for ch, c in second] == [(1, "sim-2", 2000), (2, "sim-1", 1000 + 4000)]
    assert client.unblock_read(2, 1) == 0
    assert [(ch, c.id) for ch, c in client.get_read_chunks()] == [(1, "sim-2"), (2, "sim-3")]
    assert abs(client.get_runtime() - 3.0) < 1e-6

    from uncalled_amd.__main__ import realtime_loop
    lines = []
    conf.realtime_mode = int(unc.RealtimePool.DEPLETE)
    client = unc.ClientSim(conf)
    client.add_fast5(str(f5))
    client.load_fast5s()
    client.run()
    pool = unc.RealtimePool(conf)
    realtime_loop(unc, conf, client, pool, sim=True, emit=lambda p: lines.append(str(p)), sleep=lambda s: None)
    ids = sorted(l.split("\t")[0] for l in lines)
    assert ids == ["sim-0", "sim-1", "sim-2", "sim-3"]
    for l in lines:      # deplete: mapped reads are ejected, unmapped ones kept
        assert ("\tej:f:" in l) == (l.split("\t")[2] != "*") and (("\tkp:f:" in l) or ("\ten:f:" in l) or ("\tej:f:" in l))


def case_map_pool_pipeline(unc, po, tmp_path, goldens):
    """MapPool (loader thread -> page-locked staging buffers -> mapper thread -> update()) over several fast5 files and
    batches: every read comes out once, PAF columns as the oracle maps the same signal."""
    off = goldens["sim_offsets"]
    n = 7
    reads = [dict(id="sim-%d" % i, channel=1 + i, number=i, start=100 * i, range=CAL_RANGE, offset=CAL_OFFSET, digitisation=CAL_DIGITISATION,
                  signal=goldens["sim_signal"][int(off[i]):int(off[i + 1])].tolist()) for i in range(n)]
    assert unc.write_fast5(str(tmp_path / "a.fast5"), reads[:4], True, 4000.0)
    assert unc.write_fast5(str(tmp_path / "b.fast5"), reads[4:6], True, 4000.0)
    assert unc.write_fast5(str(tmp_path / "c.fast5"), reads[6:], True, 4000.0)
    pool = unc.MapPool(_conf(unc, 512, batch_reads=3))          # 3 + 3, later 1
    pool.add_fast5(str(tmp_path / "a.fast5"))
    pool.add_fast5(str(tmp_path / "b.fast5"))
    assert pool.running()
    lines = []
    import time
    t0 = time.time()
    while pool.running():
        lines += [str(p) for p in pool.update()]
        time.sleep(0.01)
        assert time.time() - t0 < 600
    assert len(lines) == 6
    # the worker threads live until stop() (map_pool.cpp:31-42,83-97): a file added after the pool ran dry is mapped too
    pool.add_fast5(str(tmp_path / "c.fast5"))
    assert pool.running()
    while pool.running():
        lines += [str(p) for p in pool.update()]
        time.sleep(0.01)
        assert time.time() - t0 < 600
    pool.stop()
    assert not pool.running() and pool.update() == []
    oix = po.Index(G / "example_index" / "example_ref")
    om = po.Mapper(oix)
    got = {l.split("\t")[0]: l.split("\t") for l in lines}
    assert sorted(got) == sorted(r["id"] for r in reads) and len(lines) == n
    for i, r in enumerate(reads):
        o = om.map_read(po.calibrate(np.array(r["signal"], dtype=np.int16), CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
        want = po.hit_paf_cols(o, oix.ref_names())
        c = got[r["id"]]
        if o["mapped"]:
            assert (int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), int(c[7]), int(c[8]), int(c[9]), int(c[10]), int(c[11])) == want
        else:
            assert int(c[1]) == want[0] and c[2] == "*"
        assert "ch:i:%d" % r["channel"] in c and "st:i:%d" % r["start"] in c


def case_map_pool_short_of_staging_memory(unc, po, tmp_path, goldens, monkeypatch):
    """The loader with a staging buffer that cannot grow (round-3 advice: a read dropped without a PAF line, and a failed first
    read ending the run).  UNC_STAGING_MAX_KB caps a buffer at 40 KB = 20 000 samples: a batch that is full is handed over and the
    read that did not fit OPENS THE NEXT ONE; a read larger than the cap gets its unmapped line (length column as the reference
    prints it) and the run goes on.  Every read comes out exactly once, the staged ones with the oracle's columns."""
    import time
    monkeypatch.setenv("UNC_STAGING_MAX_KB", "40")
    off = goldens["sim_offsets"]
    n = 6
    reads = [dict(id="sim-%d" % i, channel=1 + i, number=i, start=100 * i, range=CAL_RANGE, offset=CAL_OFFSET, digitisation=CAL_DIGITISATION,
                  signal=goldens["sim_signal"][int(off[i]):int(off[i + 1])].tolist()[:9000]) for i in range(n)]
    big = dict(reads[2], id="too-big", signal=(reads[2]["signal"] * 4)[:30000])      # 60 KB: no buffer will ever hold it
    order = [big] + reads[:3] + [dict(big, id="too-big-2")] + reads[3:]                # ... also as the very first read of the run
    assert unc.write_fast5(str(tmp_path / "a.fast5"), order, True, 4000.0)
    pool = unc.MapPool(_conf(unc, 512, batch_reads=8))
    pool.add_fast5(str(tmp_path / "a.fast5"))
    lines, t0 = [], time.time()
    while pool.running():
        lines += [str(p) for p in pool.update()]
        time.sleep(0.01)
        assert time.time() - t0 < 600
    pool.stop()
    got = {l.split("\t")[0]: l.split("\t") for l in lines}
    assert len(lines) == len(order) and sorted(got) == sorted(r["id"] for r in order)
    for name in ("too-big", "too-big-2"):
        assert got[name][2] == "*" and int(got[name][1]) == int(np.float32(30000) * (np.float32(450.0) / np.float32(4000.0)))
    oix = po.Index(G / "example_index" / "example_ref")
    om = po.Mapper(oix)
    for r in reads:
        o = om.map_read(po.calibrate(np.array(r["signal"], dtype=np.int16), CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
        want = po.hit_paf_cols(o, oix.ref_names())
        c = got[r["id"]]
        if o["mapped"]:
            assert (int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), int(c[7]), int(c[8]), int(c[9]), int(c[10]), int(c[11])) == want
        else:
            assert int(c[1]) == want[0] and c[2] == "*"


# ---- the cases above on the lanesim build of the host module (tests/test_gpu_host.py runs them on the real one)
@pytest.mark.lanesim
def test_chunk_class(sim_host):
    case_chunk_class(sim_host)


@pytest.mark.lanesim
def test_realtime_pool_ordered_replay_matches_oracle(sim_host, oracle_lib, example, goldens):
    case_realtime_pool_ordered_replay(sim_host, oracle_lib, example, goldens, n_reads=5)     # 9 on the GPU


@pytest.mark.lanesim
def test_add_chunk_resets_the_previous_read(sim_host, oracle_lib, goldens):
    case_add_chunk_resets_the_previous_read(sim_host, oracle_lib, goldens)


@pytest.mark.lanesim
def test_oversized_chunk_is_refused(sim_host):
    case_oversized_chunk_is_refused(sim_host)


@pytest.mark.lanesim
def test_client_sim_feeds_the_decision_loop(sim_host, tmp_path, goldens):
    case_client_sim_feeds_the_decision_loop(sim_host, tmp_path, goldens)


@pytest.mark.lanesim
def test_map_pool_pipeline_matches_oracle(sim_host, oracle_lib, tmp_path, goldens):
    case_map_pool_pipeline(sim_host, oracle_lib, tmp_path, goldens)


@pytest.mark.lanesim
def test_map_pool_short_of_staging_memory(sim_host, oracle_lib, tmp_path, goldens, monkeypatch):
    case_map_pool_short_of_staging_memory(sim_host, oracle_lib, tmp_path, goldens, monkeypatch)
