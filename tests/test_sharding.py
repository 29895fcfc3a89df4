"""CPU: the N>1 path.  Reads shard across ranks with no data-path collective; covered with world_size-2 gloo
processes that each map their shard (kernel sources under the lanesim emulator) and compare the gathered
result with a single-rank run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_bounds_cover_all_reads_once():
    from uncalled_amd.sharding import shard_bounds
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 7, 100):
        off = np.concatenate(([0], np.cumsum(rng.integers(1, 50000, n)))).astype(np.uint64)
        for w in (1, 2, 3, 8):
            b = shard_bounds(off, w)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
            if n >= 4 * w:   # balanced by samples
                per = np.array([off[b[r + 1]] - off[b[r]] for r in range(w)], dtype=np.float64)
                assert per.max() <= per.mean() + 50000


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from uncalled_amd import capi
    from uncalled_amd.sharding import shard
    lib = capi.load(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so")
    gold = np.load(ROOT / "tests" / "golden" / "ref_goldens.npz")
    n = 8
    off = gold["sim_offsets"][:n + 1]
    a, b = shard(off, rank, world)
    ix = capi.Index(ROOT / "tests" / "golden" / "example_index" / "example_ref", lib=lib)   # index replicated per rank
    m = capi.Mapper(ix, n_slots=2)
    loc_off = (off[a:b + 1] - off[a]).astype(np.uint64)
    raw = gold["sim_signal"][int(off[a]):int(off[b])]
    hits = m.map_batch(raw, loc_off, capi.make_calib(b - a, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)) if b > a else \
        np.zeros(0, dtype=capi.HIT)
    np.save(Path(out_dir) / f"hits_{rank}.npy", hits)
    np.save(Path(out_dir) / f"range_{rank}.npy", np.array([a, b]))
    dist.barrier()   # the only collective: control plane, as in bench.py
    dist.destroy_process_group()


def test_two_rank_sharded_mapping_matches_single_rank(sim_lib, goldens, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from uncalled_amd import capi
    n = 8
    off = goldens["sim_offsets"][:n + 1].copy()
    ix = capi.Index(ROOT / "tests" / "golden" / "example_index" / "example_ref", lib=sim_lib)
    single = capi.Mapper(ix, n_slots=2).map_batch(goldens["sim_signal"][:int(off[n])], off,
                                                  capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
    parts = [np.load(tmp_path / f"hits_{r}.npy") for r in range(2)]
    ranges = [np.load(tmp_path / f"range_{r}.npy") for r in range(2)]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n
    assert 0 < ranges[0][1] < n
    merged = np.concatenate(parts)
    for f in ("mapped", "fwd", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "matches", "event_i", "n_nbr", "n_sa", "n_lf"):
        assert np.array_equal(merged[f], single[f]), f
