"""`-m gpu`: BASELINE config 4's reference scale.  `grch38_syn` (24 contigs, 3.1 Gbp, 30 % masked: seq_len 6.2 G, i.e.
FM rows and SA values past 2^32 -- where the reference switches to u64 ranges, src/range.hpp:37) is built on the GPU with
the chunked suffix sorter, parameterised (`uncalled index`), and a batch of reads is mapped by the HIP path and by the
oracle restatement (all host threads): PAF, winning cluster, event counts and work counters bit-exact.  At this size the
index itself selects what the small tests can only force: the natural 128-bit sort keys (no packing fits), seed-cluster
sets of hundreds of thousands of clusters drawn from the leaf pool, the 50 GB dense SA.
The index lands in bench.py's cache directory, so a bench run on the same box reuses it."""
import os
import time
from pathlib import Path

import numpy as np
import pytest

from tests.helpers import assert_hits_equal
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from uncalled_amd import capi

pytestmark = pytest.mark.gpu

N_READS = 96


@pytest.fixture(scope="module")
def grch38():
    import bench
    cache = Path(os.environ.get("UNC_BENCH_CACHE", "/tmp/uncalled_amd_bench"))
    t0 = time.time()
    prefix, codes, lens = bench.ensure_index(cache, 0, lambda: None, "grch38", "cuda:0")
    print(f"grch38_syn index ready in {time.time() - t0:.0f} s")
    return dict(prefix=prefix, codes=codes, lens=lens)


def test_grch38_scale_batch(hip_lib, oracle_lib, grch38):
    from tools.simulate_reads_torch import simulate_reads_torch
    sim = simulate_reads_torch(grch38["codes"], grch38["lens"], N_READS, seed=44, device="cuda:0")
    raw = sim["signal"].cpu().numpy()
    off = sim["offsets"]
    cal = capi.make_calib(N_READS, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    ix = capi.Index(grch38["prefix"], lib=hip_lib)
    assert ix.size == 2 * 3100000000 and ix.size > (1 << 32)
    m = capi.Mapper(ix)
    geo = m.geometry()
    t0 = time.time()
    hits = m.map_batch(raw, off, cal)
    t_gpu = time.time() - t0
    oix = oracle_lib.Index(grch38["prefix"])
    sig = oracle_lib.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    threads = max(1, min(N_READS, len(os.sched_getaffinity(0))))
    want, t_cpu = oracle_lib.map_batch(oix, sig, off, threads)
    assert_hits_equal(hits, want, "grch38")
    mapped = hits["mapped"] != 0
    assert mapped.sum() >= 0.5 * N_READS
    # the u64 paths were really taken: clusters end past row 2^32 of the FM index, both strands are hit
    assert (hits["cl_ref_en_end"][mapped] > (1 << 32)).any() and (hits["cl_ref_en_end"][mapped] < (1 << 32)).any()
    ok = 0
    for i in np.flatnonzero(mapped):
        if sim["contig"][i] >= 0 and int(hits["rid"][i]) == int(sim["contig"][i]) and bool(hits["fwd"][i]) == (sim["strand"][i] == 0) and \
                sim["pos"][i] - 100 <= hits["rf_st"][i] <= sim["pos"][i] + 3700:
            ok += 1
    assert ok >= 0.9 * mapped.sum()
    print(f"grch38 batch: {N_READS} reads, {int(mapped.sum())} mapped ({ok} at the simulated locus), gpu {t_gpu:.2f} s, "
          f"oracle on {threads} threads {t_cpu:.1f} s, mapper geometry {geo}")
