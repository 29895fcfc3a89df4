"""`-m gpu`: the reference-shaped host surface end to end -- fast5 files -> Fast5Reader -> MapPool (batches through the C
ABI, HIP kernels) -> Paf text, and the `python -m uncalled_amd map` CLI -- against the reference's own answers
(tests/golden/ref_goldens.npz; the example line is SURVEY.md's probe of `uncalled map`)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"
PREFIX = G / "example_index" / "example_ref"
EXAMPLE_COLS = ("f41a60f7-de4a-4b17-9f54-387e52d60b65\t106\t73\t106\t-\tEscherichia_coli_chromosome:2400000-2410000\t10000\t6938\t6976\t38\t39\t255"
                "\tch:i:486\tst:i:257117\tmt:f:")


@pytest.fixture(scope="module")
def unc():
    import torch
    assert torch.cuda.is_available()
    from uncalled_amd import _uncalled_amd   # fails loudly when build() has not produced it
    return _uncalled_amd


def _conf(unc, **kw):
    c = unc.Conf()
    c.bwa_prefix = str(PREFIX)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _run(pool):
    out = []
    while pool.running():
        out += pool.update()
    pool.stop()
    return out


def test_map_pool_example_fast5(unc):
    pool = unc.MapPool(_conf(unc))
    pool.add_fast5(str(G / "example_read.fast5"))
    pafs = _run(pool)
    assert len(pafs) == 1 and pafs[0].is_mapped()
    assert str(pafs[0]).startswith(EXAMPLE_COLS)
    assert float(str(pafs[0]).rsplit(":", 1)[1]) > 0
    assert not pool.running() and pool.update() == []


def _sim_reads(gold):
    off = gold["sim_offsets"].astype(np.int64)
    return [dict(id="sim-%04d" % i, channel=1 + i % 512, number=i, start=100 * i, range=1534.14, offset=10.0, digitisation=8192.0,
                 signal=gold["sim_signal"][off[i]:off[i + 1]].tolist()) for i in range(off.size - 1)]


def _check_against_golden(lines, gold):
    f = {str(n): j for j, n in enumerate(gold["hit_fields"])}
    by_id = {l.split("\t")[0]: l.split("\t") for l in lines}
    assert len(by_id) == gold["sim_hits"].shape[0]
    n_mapped = 0
    for i, want in enumerate(gold["sim_hits"]):
        c = by_id["sim-%04d" % i]
        assert int(c[1]) == want[f["rd_len"]]
        assert c[-3:-1] == ["ch:i:%d" % (1 + i % 512), "st:i:%d" % (100 * i)] and c[-1].startswith("mt:f:")
        if not want[f["mapped"]]:
            assert c[2:12] == ["*"] * 9 + ["255"]
            continue
        n_mapped += 1
        assert [int(c[2]), int(c[3]), c[4]] == [want[f["rd_st"]], want[f["rd_en"]], "+" if want[f["fwd"]] else "-"]
        assert [int(c[6]), int(c[7]), int(c[8]), int(c[9])] == [want[f["rf_len"]], want[f["rf_st"]], want[f["rf_en"]], want[f["matches"]]]
        assert int(c[10]) == want[f["rf_en"]] - want[f["rf_st"]] + 1 and c[11] == "255"
    assert n_mapped > 20


def test_map_pool_multi_fast5_batches(unc, tmp_path):
    gold = np.load(G / "ref_goldens.npz")
    reads = _sim_reads(gold)
    unc.write_fast5(str(tmp_path / "a.fast5"), reads[:30], True)
    unc.write_fast5(str(tmp_path / "b.fast5"), reads[30:], True)
    pool = unc.MapPool(_conf(unc, batch_reads=20))   # three batches: 20 + 20 + 8
    pool.add_fast5(str(tmp_path / "a.fast5"))
    pool.add_fast5(str(tmp_path / "b.fast5"))
    sizes, lines = [], []
    while pool.running():
        got = pool.update()
        sizes.append(len(got))
        lines += [str(p) for p in got]
    # update() never blocks: whole batches arrive, possibly several at once, with empty polls in between
    got_sizes = [x for x in sizes if x]
    assert sum(got_sizes) == 48 and all(x in (8, 20, 28, 40, 48) for x in got_sizes) and pool.batch_reads() == 20
    _check_against_golden(lines, gold)


def test_cli_map_and_pafstats(unc, tmp_path):
    gold = np.load(G / "ref_goldens.npz")
    reads = _sim_reads(gold)
    d = tmp_path / "fast5"
    (d / "sub").mkdir(parents=True)
    unc.write_fast5(str(d / "a.fast5"), reads[:24], True)
    unc.write_fast5(str(d / "sub" / "b.fast5"), reads[24:], True)
    (d / "notes.txt").write_text("not a fast5\n")
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "map", str(PREFIX), str(d), "-r", "-t", "16"], cwd=str(ROOT), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Loading fast5s" in r.stderr and "Finishing" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l]
    _check_against_golden(lines, gold)
    # read-id list + max reads through the CLI, then pafstats on the output
    ids = tmp_path / "ids.txt"
    ids.write_text("sim-0003\nsim-0030\nsim-0041\n")
    r2 = subprocess.run([sys.executable, "-m", "uncalled_amd", "map", str(PREFIX), str(d), "-r", "-l", str(ids)], cwd=str(ROOT), capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert sorted(l.split("\t")[0] for l in r2.stdout.splitlines() if l) == ["sim-0003", "sim-0030", "sim-0041"]
    paf = tmp_path / "out.paf"
    paf.write_text(r.stdout)
    r3 = subprocess.run([sys.executable, "-m", "uncalled_amd", "pafstats", str(paf), "-r", str(paf)], cwd=str(ROOT), capture_output=True, text=True,
                        timeout=120)
    assert r3.returncode == 0, r3.stderr[-2000:]
    n_mapped = int(gold["sim_hits"][:, 0].sum())
    assert r3.stdout.startswith("Summary: 48 reads, %d mapped" % n_mapped)
    assert "BP per sec:" in r3.stdout


def test_missing_index_is_fatal(unc, tmp_path):
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "map", str(tmp_path / "nothing"), str(G / "example_read.fast5")], cwd=str(ROOT),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "does not exist" in r.stderr


def test_cli_index_reproduces_bundled_uncl(unc, tmp_path):
    """`python -m uncalled_amd index` on the bundled 10 kb reference: BWA-format files byte-identical to the bundled
    `bwa index` output and a .uncl identical to the one the reference shipped (scripts/uncalled:38-78)."""
    import shutil
    ex = G / "example_index"
    fa = tmp_path / "example_ref.fa"
    shutil.copyfile(ex / "example_ref.fa", fa)
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "index", str(fa)], cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for suf in (".amb", ".ann", ".bwt", ".pac", ".sa"):
        assert (tmp_path / ("example_ref.fa" + suf)).read_bytes() == (ex / ("example_ref" + suf)).read_bytes(), suf
    assert (tmp_path / "example_ref.fa.uncl").read_text() == (ex / "example_ref.uncl").read_text()
    # a second run keeps the BWA files and only redoes the parameter search
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "index", str(fa), "--speeds", "50"], cwd=str(ROOT), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "Using previously built BWA index" in r.stderr
    names = [l.split("\t")[0] for l in (tmp_path / "example_ref.fa.uncl").read_text().splitlines()]
    assert names == ["default", "speed_50"]


# ---- realtime host classes + the pipelined MapPool on the real GPU (the same cases tests/test_realtime_host.py runs under lanesim)
def test_realtime_pool_ordered_replay_gpu(unc, oracle_lib, example, goldens):
    from tests.test_realtime_host import case_chunk_class, case_realtime_pool_ordered_replay
    case_chunk_class(unc)
    case_realtime_pool_ordered_replay(unc, oracle_lib, example, goldens)


def test_realtime_add_chunk_and_decision_loop_gpu(unc, oracle_lib, tmp_path, goldens):
    from tests.test_realtime_host import (case_add_chunk_resets_the_previous_read, case_client_sim_feeds_the_decision_loop,
                                          case_oversized_chunk_is_refused)
    case_add_chunk_resets_the_previous_read(unc, oracle_lib, goldens)
    case_oversized_chunk_is_refused(unc)
    case_client_sim_feeds_the_decision_loop(unc, tmp_path, goldens)


def test_map_pool_pipeline_gpu(unc, oracle_lib, tmp_path, goldens):
    from tests.test_realtime_host import case_map_pool_pipeline
    case_map_pool_pipeline(unc, oracle_lib, tmp_path, goldens)


def test_map_pool_short_of_staging_memory_gpu(unc, oracle_lib, tmp_path, goldens, monkeypatch):
    from tests.test_realtime_host import case_map_pool_short_of_staging_memory
    case_map_pool_short_of_staging_memory(unc, oracle_lib, tmp_path, goldens, monkeypatch)
