#!/usr/bin/env python3
"""Dev checker (CPU): the parametrisations that only the GPU suite runs (tests/test_gpu_parity.py: more reads, more channels,
larger cases than the emulator suite affords on every run), on the lanesim build of the kernel sources.  For the times the GPU
is not at hand and the round-end run of `pytest -m gpu` is the first one to see a change: half an hour on one core per group.

    python tests/dev/gpu_params_on_emulator.py [chunked] [mid] [pool] [batch] [t1] [sliced]        (default: all)

Round 4: a slot state written without its node count leaked a pool chunk per decided read; the emulator suite's chunked cases
(at most six reads per channel) passed, (3 channels, 31 reads) went dry in round 28."""
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402
from uncalled_amd import capi  # noqa: E402
import parity_cases as pc  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main():
    groups = sys.argv[1:] or ["chunked", "mid", "pool", "batch", "t1", "sliced"]
    ex = np.load(GOLD / "example_read.npz")
    example = dict(signal=ex["signal"], range=float(ex["range"]), offset=float(ex["offset"]), digitisation=float(ex["digitisation"]),
                   prefix=GOLD / "example_index" / "example_ref")
    goldens = np.load(GOLD / "ref_goldens.npz")
    lib = capi.load(os.environ.get("UNC_FUZZ_LIB") or (ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"))
    t0 = time.time()

    def ok(what):
        print(f"{what}: ok [{time.time() - t0:.0f} s]", flush=True)

    def with_env(env, fn):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            fn()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    for g in groups:
        if g == "chunked":          # test_chunked_realtime_path
            for a in [(1, 31, None, False), (3, 31, None, False), (2, 8, 2, False), (2, 6, None, True)]:
                with_env({"UNC_RT_TEAM": os.environ.get("UNC_RT_TEAM", "2")}, lambda: pc.case_chunked_realtime_path(lib, pyoracle, example, goldens, *a))
                ok(f"chunked {a}")
        elif g == "mid":            # test_chunked_mid_reference_team_sort
            for team in (2, 8):
                with tempfile.TemporaryDirectory() as t:
                    with_env({"UNC_RT_TEAM": str(team)}, lambda: pc.case_chunked_mid_reference(lib, pyoracle, Path(t), n=4, cut=8000))
                ok(f"chunked mid reference, team {team}")
        elif g == "pool":           # test_chunked_pool_chunks_go_back
            for team in (1, 8):
                with_env({"UNC_RT_TEAM": str(team), "UNC_RT_POOL_CHUNKS": "2"},
                         lambda: pc.case_chunked_realtime_path(lib, pyoracle, example, goldens, 1, 12, None))
                ok(f"pool of two chunks, 12 reads, team {team}")
        elif g == "batch":          # test_synthetic_batch
            for a in [(300, 24), (97, 12), (10000, 48)]:
                pc.case_synthetic_batch(lib, pyoracle, example, goldens, *a)
                ok(f"batch {a}")
        elif g == "t1":             # test_read_order_t1
            for a in [(97, 24, 7), (130, 32, 20), (200, 48, 1)]:
                pc.case_read_order_t1(lib, pyoracle, example, goldens, *a)
                ok(f"-t 1 order {a}")
        elif g == "sliced":         # test_sliced_scheduler, test_cluster_pool_pressure
            for a in [(10000, 37, 5, 2), (300, 11, 3, 1), (10000, 200, 9, 4)]:
                pc.case_sliced_scheduler(lib, pyoracle, example, goldens, *a)
                ok(f"sliced {a}")
            for a in [(3, 2), (1, 1)]:
                pc.case_cluster_pool_pressure(lib, pyoracle, example, goldens, *a)
                ok(f"pool pressure {a}")
        else:
            print(f"unknown group {g}")
            return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
