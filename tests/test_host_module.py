"""The pybind11 host surface (`_uncalled_amd`: Conf / Fast5Reader / ReadBuffer / Paf / MapPool) -- CPU part: the fast5
reader against the bundled example read, both fast5 layouts, the read-id filter / max_reads / max_chunks rules of
fast5_reader.cpp:62-248 and read_buffer.cpp:198-246, and the PAF text of read_buffer.cpp:96-137."""
from pathlib import Path

import numpy as np
import pytest

unc = pytest.importorskip("uncalled_amd._uncalled_amd", reason="run __graft_entry__.build() first")

G = Path(__file__).resolve().parent / "golden"


def _reads(n, seed=0):
    rng = np.random.default_rng(seed)
    return [dict(id="%08x-sim" % (i * 2654435761 % 2 ** 32), channel=1 + (i * 37) % 512, number=i, start=4000 * i, range=1400.0 + i,
                 offset=float(i % 7), digitisation=8192.0, signal=rng.integers(150, 1100, 4100 + 13 * i).astype(np.int16).tolist())
            for i in range(n)]


def _drain(r):
    out = []
    while not r.empty():
        if r.buffer_size() == 0 and r.fill_buffer() == 0:
            break
        out.append(r.pop_read())
    return out


def test_example_fast5_matches_fixture():
    ex = np.load(G / "example_read.npz")
    r = unc.Fast5Reader("", "", 0, 100)
    r.add_fast5(str(G / "example_read.fast5"))
    assert r.fill_buffer() == 1
    rd = r.pop_read()
    assert r.empty()
    assert np.array_equal(np.array(rd.raw_i16, dtype=np.int16), ex["signal"])
    rng, off, dig = rd.calibration
    # 1534.141357421875 in the file, "1534.14" after the reference's string round trip (read_buffer.cpp:213-222)
    assert (np.float32(rng), off, dig) == (np.float32(ex["range"]), float(ex["offset"]), float(ex["digitisation"]))
    assert rd.channel == 486 and rd.id == "f41a60f7-de4a-4b17-9f54-387e52d60b65" and rd.size() == ex["signal"].size
    gold = np.load(G / "ref_goldens.npz")
    if "ex_signal" in gold:
        assert np.array_equal(np.array(rd.raw, dtype=np.float32), gold["ex_signal"])


def test_multi_and_single_layouts_roundtrip(tmp_path):
    reads = _reads(7)
    assert unc.write_fast5(str(tmp_path / "m.fast5"), reads[:6], True)
    assert unc.write_fast5(str(tmp_path / "s.fast5"), reads[6:], False)
    assert not unc.write_fast5(str(tmp_path / "bad.fast5"), reads[:2], False)
    r = unc.Fast5Reader("", "", 0, 3)   # small buffer: fill_buffer is called repeatedly
    r.add_fast5(str(tmp_path / "m.fast5"))
    r.add_fast5(str(tmp_path / "s.fast5"))
    got = _drain(r)
    assert len(got) == 7
    by_id = {d["id"]: d for d in reads}
    for g in got:
        d = by_id[g.id]
        assert g.raw_i16 == d["signal"] and g.channel == d["channel"] and g.start == d["start"] and g.number == d["number"]
        assert g.calibration == (float(np.float32(d["range"])), d["offset"], d["digitisation"])
        cal = (np.array(d["signal"], dtype=np.uint16).astype(np.float32) + np.float32(d["offset"])) * np.float32(d["range"]) / np.float32(8192)
        assert np.array_equal(np.array(g.raw, dtype=np.float32), cal.astype(np.float32))
    assert got[-1].id == reads[6]["id"]   # files are consumed in the order they were added


def test_read_filter_max_reads_and_lists(tmp_path):
    reads = _reads(10, seed=3)
    unc.write_fast5(str(tmp_path / "a.fast5"), reads[:5], True)
    unc.write_fast5(str(tmp_path / "b.fast5"), reads[5:], True)
    (tmp_path / "fast5s.txt").write_text("%s\n%s\n" % (tmp_path / "a.fast5", tmp_path / "b.fast5"))
    want = [reads[1]["id"], reads[7]["id"], reads[8]["id"]]
    (tmp_path / "reads.txt").write_text("\n".join(want) + "\n")
    r = unc.Fast5Reader(str(tmp_path / "fast5s.txt"), str(tmp_path / "reads.txt"), 0, 100)
    assert sorted(x.id for x in _drain(r)) == sorted(want)
    # max_reads caps the filter itself (fast5_reader.cpp:88-96) and the number of reads buffered
    r = unc.Fast5Reader(str(tmp_path / "fast5s.txt"), str(tmp_path / "reads.txt"), 2, 100)
    assert sorted(x.id for x in _drain(r)) == sorted(want[:2])
    r = unc.Fast5Reader(str(tmp_path / "fast5s.txt"), "", 4, 100)
    assert len(_drain(r)) == 4 and r.all_buffered()
    r = unc.Fast5Reader("", "", 0, 100)
    assert not r.load_fast5_list(str(tmp_path / "missing.txt")) and r.empty()
    r.add_fast5(str(tmp_path / "nope.fast5"))          # unreadable file: reported, skipped
    r.add_fast5(str(tmp_path / "a.fast5"))
    assert len(_drain(r)) == 5


def test_max_chunks_truncates_signal(tmp_path):
    reads = _reads(2, seed=5)
    reads[0]["signal"] = list(range(9001))
    reads[1]["signal"] = list(range(7000))
    unc.write_fast5(str(tmp_path / "m.fast5"), reads, True)
    c = unc.Conf()
    c.max_chunks = 2
    r = unc.Fast5Reader(c)
    r.add_fast5(str(tmp_path / "m.fast5"))
    sizes = {x.id: x.size() for x in _drain(r)}
    assert sizes == {reads[0]["id"]: 8000, reads[1]["id"]: 7000}


def test_paf_text():
    p = unc.Paf()
    assert str(p) == "\t0\t*\t*\t*\t*\t*\t*\t*\t*\t*\t255"
    p.set_int(unc.Paf.EJECT, 3)
    p.set_float(unc.Paf.MAP_TIME, 12.5)
    p.set_str(unc.Paf.KEEP, "x")
    assert str(p).endswith("255\tej:i:3\tmt:f:12.500000\tkp:Z:x")
    assert not p.is_mapped() and not p.is_ended()


def test_conf_defaults_and_cli_parser():
    c = unc.Conf()
    assert (c.threads, c.idx_preset, c.max_events, c.max_chunks, c.num_channels, c.chunk_time) == (1, "default", 30000, 1000000, 512, 1.0)
    from uncalled_amd.__main__ import get_parser
    a = get_parser().parse_args(["map", "ref/prefix", "dir1", "x.fast5", "-n", "5", "-e", "100", "-l", "ids.txt", "-t", "8"])
    assert (a.bwa_prefix, a.fast5s, a.max_reads, a.max_events, a.read_list, a.threads) == ("ref/prefix", ["dir1", "x.fast5"], 5, 100, "ids.txt", 8)
    a = get_parser().parse_args(["index", "g.fa", "--probs", "0.1,0.2"])
    assert a.bwa_prefix is None and a.probs == "0.1,0.2" and a.max_sample_dist == 100


def test_multi_gpu_file_sharding():
    from uncalled_amd.__main__ import get_parser, shard_files
    files = ["f%02d.fast5" % i for i in range(11)] + [None]
    shards = shard_files(files, 4)
    assert [len(x) for x in shards] == [3, 3, 3, 2] and sorted(sum(shards, [])) == files[:-1]
    assert shards[1] == ["f01.fast5", "f05.fast5", "f09.fast5"]
    assert shard_files(["a"], 8)[0] == ["a"] and all(not x for x in shard_files(["a"], 8)[1:])
    a = get_parser().parse_args(["map", "ref", "reads/", "--gpus", "8"])
    assert a.gpus == 8 and a.device == 0


def test_multi_gpu_launcher_reports_failed_workers(tmp_path):
    """No GPU here, so both workers die in MapPool's constructor: the launcher must say so and exit non-zero."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "map", str(G / "example_index" / "example_ref"), str(G / "example_read.fast5"),
                        str(G / "example_read.fast5"), "--gpus", "2"], cwd=str(root), capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and len([l for l in r.stdout.splitlines() if l]) == 2
    elif not torch.cuda.is_available():
        assert r.returncode == 1 and "worker exit codes" in r.stderr and r.stdout == ""


def test_multi_gpu_max_reads_and_read_list(tmp_path, capsys):
    """`map --gpus 2 -n K -l list`: every worker gets the whole read list and no `-n`; the cap is applied once, by the
    launcher, on the forwarded PAF lines (per-worker cuts mapped at most K/N reads)."""
    import glob
    import sys
    import tempfile
    from uncalled_amd.__main__ import get_parser, map_multi_gpu, worker_cmd
    files = []
    for i in range(4):
        f = tmp_path / ("r%d.fast5" % i)
        f.write_bytes(b"")
        files.append(str(f))
    ids = tmp_path / "ids.txt"
    ids.write_text("a\nb\nc\n")
    a = get_parser().parse_args(["map", "ref", *files, "--gpus", "2", "-n", "5", "-l", str(ids)])
    cmd = worker_cmd(a, "list.txt", 1)
    assert "-n" not in cmd and cmd[cmd.index("-l") + 1] == str(ids) and cmd[cmd.index("--device") + 1] == "1"
    before = set(glob.glob(tempfile.gettempdir() + "/*.fast5s.txt"))

    def fake(args, list_name, dev):   # a worker that "maps" four reads per fast5 file of its shard
        return [sys.executable, "-c", "import sys\nfor f in open(sys.argv[1]):\n  [print('read-%s-%d' % (f.strip()[-8:], j), flush=True) for j in range(4)]", list_name]

    map_multi_gpu(a, None, make_cmd=fake)
    out = [l for l in capsys.readouterr().out.splitlines() if l]
    assert len(out) == 5 and len(set(out)) == 5                      # exactly -n lines of the 16 produced
    a.max_reads = None
    map_multi_gpu(a, None, make_cmd=fake)
    assert len([l for l in capsys.readouterr().out.splitlines() if l]) == 16
    assert set(glob.glob(tempfile.gettempdir() + "/*.fast5s.txt")) == before   # list files are removed
