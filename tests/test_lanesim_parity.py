"""The product's HIP kernel sources, compiled unmodified against the lanesim CPU SIMT emulator (tests/lanesim),
checked against the oracle.  Logic tests for the GPU-less build container; the same cases run on a real MI355X
through the gfx950 library in test_gpu_parity.py."""
import pytest

from tests import parity_cases as pc

pytestmark = pytest.mark.lanesim


def test_index_tables(sim_lib, oracle_lib, example, goldens):
    pc.case_index_tables(sim_lib, oracle_lib, example, goldens)


def test_fm_primitives(sim_lib, oracle_lib, example, goldens):
    pc.case_fm_primitives(sim_lib, oracle_lib, example, goldens)


def test_events_and_normaliser(sim_lib, oracle_lib, example, goldens):
    pc.case_events_and_normaliser(sim_lib, oracle_lib, example, goldens)


def test_radix_sort_of_the_index_builders(sim_lib):
    pc.case_radix_sort(sim_lib)


def test_calib_chase_runs_over_a_region(sim_lib, example):
    """unc_calib_chase (diagnostics: dependent loads over a region of device memory): two wavefronts, a handful of steps over a mapper's
    own slots -- the kernel stays inside the region whatever the region holds, and the entry refuses a region of less than one load."""
    import ctypes as C
    import numpy as np
    from uncalled_amd import capi
    ix = capi.Index(example["prefix"], lib=sim_lib)
    m = capi.Mapper(ix, n_slots=2, n_waves=2)
    a = np.zeros(6, dtype=np.uint64)
    sim_lib.unc_mapper_device_addresses.argtypes = [C.c_void_p, C.c_void_p]
    sim_lib.unc_calib_chase.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    assert sim_lib.unc_mapper_device_addresses(m.h, a.ctypes.data) == 0 and int(a[0]) and int(a[1]) and int(a[2])
    ms = C.c_float(-1.0)
    assert sim_lib.unc_calib_chase(0, int(a[0]), int(a[1]) * 2, 2, 16, C.byref(ms)) == 0 and ms.value >= 0.0
    assert sim_lib.unc_calib_chase(0, int(a[0]), 8, 2, 16, C.byref(ms)) != 0


def test_suffix_array_built_behind_the_c_abi(sim_lib):
    """unc_build_suffix_array (the suffix sort of `uncalled index` without torch: k_sort.hip's radix sort and the steps between the sorts)
    against the numpy prefix doubling, on texts with ties far deeper than the first 21-symbol key."""
    import numpy as np
    from uncalled_amd import capi
    from uncalled_amd.build_index import suffix_array
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 64, 2049, 6000):
        t = rng.integers(0, 4, size=n).astype(np.uint8)
        if n == 6000:
            t[1000:3000] = 0                                                        # a homopolymer run
            t[3000:3600] = np.tile(np.array([0, 1, 2, 3], dtype=np.uint8), 150)     # a tandem repeat
            t[4000:5000] = t[200:1200]                                              # an exact repeat
        assert np.array_equal(capi.build_suffix_array(t, 0, sim_lib), suffix_array(t)), n


def test_events_of_the_reads_the_sweep_found_wrong(sim_lib, oracle_lib, example):
    pc.case_events_sweep_reads(sim_lib, oracle_lib, example)


def test_events_edge_cases(sim_lib, oracle_lib, example, goldens):
    pc.case_events_edge_cases(sim_lib, oracle_lib, example, goldens)


def test_example_read_full_path(sim_lib, oracle_lib, example, goldens):
    pc.case_example_read_full_path(sim_lib, oracle_lib, example, goldens)


def test_batch_in_two_halves(sim_lib, oracle_lib, example, goldens):
    pc.case_batch_in_two_halves(sim_lib, oracle_lib, example, goldens)


def test_same_row_two_kmers_walked_again_on_wide_keys(sim_lib, oracle_lib, tmp_path):
    pc.case_same_row_two_kmers(sim_lib, oracle_lib, tmp_path)


@pytest.mark.parametrize("max_paths,n_reads", [(10000, 6), (97, 6), (300, 8)])
def test_synthetic_batch(sim_lib, oracle_lib, example, goldens, max_paths, n_reads):
    pc.case_synthetic_batch(sim_lib, oracle_lib, example, goldens, max_paths, n_reads)


@pytest.mark.parametrize("max_paths,n_reads,split", [(97, 6, 2), (130, 6, 3)])
def test_read_order_t1(sim_lib, oracle_lib, example, goldens, max_paths, n_reads, split):
    pc.case_read_order_t1(sim_lib, oracle_lib, example, goldens, max_paths, n_reads, split)


def test_trace_matches_oracle_every_event(sim_lib, oracle_lib, example, goldens):
    pc.case_trace_matches_oracle_every_event(sim_lib, oracle_lib, example, goldens)


@pytest.mark.parametrize("n_channels,n_reads,max_chunks", [(1, 4, None), (2, 6, 2)])   # (3, 31) runs on the GPU
def test_chunked_realtime_path(sim_lib, oracle_lib, example, goldens, n_channels, n_reads, max_chunks):
    pc.case_chunked_realtime_path(sim_lib, oracle_lib, example, goldens, n_channels, n_reads, max_chunks)


@pytest.mark.parametrize("team", [2, 1])
def test_chunked_pool_chunks_go_back(sim_lib, oracle_lib, example, goldens, monkeypatch, team):
    """A channel's chunks of the node pool go back when its read is decided or replaced: six reads in a row on ONE channel with a pool
    of two chunks (UNC_RT_POOL_CHUNKS; a read here needs one) -- a chunk that is not handed back leaves the third read dry.  (Round 4:
    a slot state written without its node count did exactly that, and no chunked case of this suite was long enough to notice.)"""
    monkeypatch.setenv("UNC_RT_TEAM", str(team))
    monkeypatch.setenv("UNC_RT_POOL_CHUNKS", "2")
    pc.case_chunked_realtime_path(sim_lib, oracle_lib, example, goldens, 1, 6, None)


@pytest.mark.parametrize("team", [8, 4, 1])
def test_chunked_team_sizes(sim_lib, oracle_lib, example, goldens, monkeypatch, team):
    """k_map_team with 8 and 4 wavefronts per channel, and the one-wavefront kernel (the other chunked cases of this suite run on
    teams of 2, see conftest.py): the same reads, chunk by chunk, as the oracle maps them."""
    monkeypatch.setenv("UNC_RT_TEAM", str(team))
    pc.case_chunked_realtime_path(sim_lib, oracle_lib, example, goldens, 2, 3, None)


@pytest.mark.parametrize("shift", [4, 8])
def test_narrow_buckets(sim_lib, oracle_lib, example, goldens, monkeypatch, shift):
    pc.case_narrow_buckets(sim_lib, oracle_lib, example, goldens, monkeypatch, shift)


def test_loud_overflows(sim_lib, oracle_lib, example, goldens):
    pc.case_loud_overflows(sim_lib, oracle_lib, example, goldens)


def test_unsorted_stream_past_one_block(sim_lib, oracle_lib, tmp_path):
    pc.case_unsorted_stream_past_one_block(sim_lib, oracle_lib, tmp_path)


def test_chunked_flags_carry_over(sim_lib, oracle_lib, tmp_path):
    pc.case_chunked_flags_carry_over(sim_lib, oracle_lib, tmp_path)


def test_chunked_stage_tap(sim_lib, oracle_lib, example, goldens):
    pc.case_chunked_stage_tap(sim_lib, oracle_lib, example, goldens, n_reads=6)


def test_chunked_variants(sim_lib, oracle_lib, example, goldens):
    pc.case_chunked_variants(sim_lib, oracle_lib, example, goldens, n_channels=2, n_reads=3)     # 6 reads on the GPU


def test_parameter_variants(sim_lib, oracle_lib, example, goldens):
    pc.case_parameter_variants(sim_lib, oracle_lib, example, goldens, n=3)       # 6 reads per set on the GPU


def test_cluster_overflow_remap(sim_lib, oracle_lib, example, goldens):
    pc.case_cluster_overflow_remap(sim_lib, oracle_lib, example, goldens)


def test_wide_sort_keys(sim_lib, oracle_lib, example, goldens, monkeypatch):
    pc.case_wide_sort_keys(sim_lib, oracle_lib, example, goldens, monkeypatch, n=6)       # 10 reads on the GPU


@pytest.mark.parametrize("max_paths,slice_events,n_slots,n_waves", [(10000, 37, 5, 2), (300, 11, 3, 1)])
def test_sliced_scheduler(sim_lib, oracle_lib, example, goldens, max_paths, slice_events, n_slots, n_waves):
    pc.case_sliced_scheduler(sim_lib, oracle_lib, example, goldens, max_paths, slice_events, n_slots, n_waves, n_reads=6)


def test_scheduler_rings_per_xcd(sim_lib, oracle_lib, example, goldens):
    assert pc.case_scheduler_rings_per_xcd(sim_lib, oracle_lib, example, goldens, n_reads=12) == 8


@pytest.mark.parametrize("pool_chunks,n_waves", [(1, 1)])
def test_cluster_pool_pressure(sim_lib, oracle_lib, example, goldens, pool_chunks, n_waves):
    pc.case_cluster_pool_pressure(sim_lib, oracle_lib, example, goldens, pool_chunks, n_waves, n_reads=4)


def test_big_forests(sim_lib, oracle_lib, example, goldens, tmp_path, monkeypatch):
    pc.case_big_forests(sim_lib, oracle_lib, example, goldens, tmp_path, monkeypatch)


@pytest.fixture(scope="module")
def sim_lib_norepair():
    """The emulator library built with UNC_MERGE_REPAIR=0: the runs of child keys reach the merge with their out-of-order
    pairs still in them, so the merge's own check has to notice and send the event through the bitonic network."""
    from pathlib import Path
    from uncalled_amd import capi
    from conftest import locked_make
    root = Path(__file__).resolve().parents[1]
    locked_make("-C", str(root / "tests" / "lanesim"), "OUT=_build_norepair", "EXTRA=-DUNC_MERGE_REPAIR=0")
    return capi.load(root / "tests" / "lanesim" / "_build_norepair" / "libuncalled_sim.so")


def test_merge_check_and_fallback(sim_lib_norepair, oracle_lib, example, goldens, tmp_path, monkeypatch):
    pc.case_big_forests(sim_lib_norepair, oracle_lib, example, goldens, tmp_path, monkeypatch, wide_too=False)


def test_merge_walk_mid_reference(sim_lib, oracle_lib, tmp_path):
    pc.case_mid_reference(sim_lib, oracle_lib, tmp_path, n=2, cut=6000)          # (3 reads of 8000 samples on the GPU)


@pytest.mark.parametrize("team,n", [(2, 2), (4, 2), (8, 2)])       # (1 as well on the GPU; a team of 8 is two minutes of emulation)
def test_chunked_mid_reference_team_sort(sim_lib, oracle_lib, tmp_path, monkeypatch, team, n):
    monkeypatch.setenv("UNC_RT_TEAM", str(team))
    pc.case_chunked_mid_reference(sim_lib, oracle_lib, tmp_path, n=n, cut=2600, chunk_len=2000, n_channels=1)     # (4 reads of 8000 samples on the GPU)


def _fuzz_round(lib_path, seed, mode, team):
    """one seeded round of tests/dev/fuzz_parity.py in a process of its own (it loads the library named by UNC_FUZZ_LIB)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, UNC_RT_TEAM=str(team))
    if lib_path:
        env["UNC_FUZZ_LIB"] = str(lib_path)
    r = subprocess.run([sys.executable, str(root / "tests" / "dev" / "fuzz_parity.py"), "1", str(seed), mode], env=env, capture_output=True, text=True,
                       timeout=900, cwd=str(root))
    assert r.returncode == 0 and "device == oracle" in r.stdout, r.stdout[-800:] + r.stderr[-400:]


def test_team_round_that_fills_the_buffer_exactly(sim_lib):
    """Found by the seeded chunked fuzz run on teams of 8 (round 4, seed 9512: max_paths 60): a round of the team's phase E whose
    children fill the path buffer EXACTLY cuts no child, yet the parents behind the last child are not reached any more
    (mapper.cpp:521-523) -- their dead-end seeds had been counted into the totals the waves publish, and one SA look-up too many
    followed.  The round now exchanges the kept counts like any round that hits the cut-off."""
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    _fuzz_round(root / "tests" / "lanesim" / "_build" / "libuncalled_sim.so", 9512, "rt", 2)
