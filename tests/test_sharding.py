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


def _split(n, rank, world):
    """contiguous share of n reads for `rank` (the product shards by file / by seed: uncalled_amd/__main__.py, bench.py)"""
    return n * rank // world, n * (rank + 1) // world


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
    from uncalled_amd import capi
    lib = capi.load(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so")
    gold = np.load(ROOT / "tests" / "golden" / "ref_goldens.npz")
    n = 6
    off = gold["sim_offsets"][:n + 1]
    a, b = _split(n, rank, world)
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
    n = 6
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


def test_bench_py_two_ranks_gloo(sim_lib, tmp_path):
    """bench.py's own world > 1 branch (rank-0 index step, barriers, MAX all-reduce of the timed region, one JSON line from
    rank 0) on two CPU ranks: UNC_DIST_BACKEND=gloo, kernels = the lanesim build of the same sources."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, UNC_DIST_BACKEND="gloo", UNC_BENCH_LIB=str(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"),
               UNC_BENCH_CACHE=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "example", "--reads", "3"], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    from test_bench_contract import strict_line
    b = strict_line(lines[0])                                       # compact, strictly parseable (round 4's line was not)
    assert b["n_gpus"] == 2 and b["steps"] == 2 and b["scaling"] == "weak" and b["metric"] == "reads_mapped_per_sec"
    assert b["config"]["reads_per_gpu_per_step"] == 3
    assert abs(b["value"] - 3 * 2 / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-4     # whole job: both ranks' reads
    assert b["verify"]["all_steps_identical"] and b["verify"]["steps_hashed"] == 2
    assert "cpu_baseline" not in b and "secondary" not in b                               # N > 1: neither is run


def test_bench_py_launches_its_own_ranks(sim_lib, tmp_path):
    """plain `python bench.py --gpus 2` (no launcher): the script starts the two ranks itself, the line says n_gpus 2, and the
    secondary block runs on both ranks as well (the N > 1 form of BASELINE config 4; here on the example index)."""
    import json
    import subprocess
    env = dict(os.environ, UNC_DIST_BACKEND="gloo", UNC_BENCH_LIB=str(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"),
               UNC_BENCH_CACHE=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "example",
                        "--reads", "2", "--secondary", "example"], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    from test_bench_contract import strict_line
    b = strict_line(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["reads_per_gpu_per_step"] == 2
    sec = b["secondary"]["example"]
    assert sec["n_gpus"] == 2 and sec["value"] > 0 and sec["verify"]["all_steps_identical"]
    assert abs(sec["value"] - 2 * 2 / (sec["ms_per_step"] * 1e-3)) / sec["value"] < 1e-4
    # a launcher that started a different number of ranks than --gpus says is refused
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "example", "--reads", "2"], env=env1,
                       capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode != 0 and "n_gpus" in r.stderr


def test_bench_py_eight_ranks_dry_run(sim_lib, tmp_path):
    """The driver's 8-GPU run, dry: plain `python bench.py --gpus 8` on eight gloo ranks over the emulator library (tiny workload).  No
    hardware curve can be measured here; what CAN be checked is that eight ranks rendezvous, rank 0 alone builds / names the index
    behind the barrier, every rank maps its own shard, the MAX-over-ranks timing and the budget all-reduce agree on eight ranks, and
    ONE compact line comes out with n_gpus 8 in the headline and in the secondary block (the N > 1 form of BASELINE config 4)."""
    import json
    import subprocess
    from test_bench_contract import strict_line
    env = dict(os.environ, UNC_DIST_BACKEND="gloo", UNC_BENCH_LIB=str(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"),
               UNC_BENCH_CACHE=str(tmp_path), UNC_BENCH_DETAIL=str(tmp_path / "bench_detail.json"), OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--workload", "example",
                        "--reads", "1", "--secondary", "example", "--secondary-steps", "1"], env=env, capture_output=True, text=True,
                       timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    b = strict_line(lines[0])
    assert b["n_gpus"] == 8 and b["scaling"] == "weak" and b["config"]["reads_per_gpu_per_step"] == 1
    assert abs(b["value"] - 8 * 1 / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-4          # whole job: eight ranks' reads
    sec = b["secondary"]["example"]
    assert sec["n_gpus"] == 8 and sec["verify"]["all_steps_identical"] and abs(sec["value"] - 8 / (sec["ms_per_step"] * 1e-3)) / sec["value"] < 1e-4
    assert "cpu_baseline" not in b                                                            # N > 1: no rank runs (or waits for) a CPU leg
    detail = json.loads((tmp_path / "bench_detail.json").read_text())
    assert detail["n_gpus"] == 8 and detail["secondary"]["example"]["n_gpus"] == 8


def test_launcher_without_gpus_flag_is_adopted(sim_lib, tmp_path):
    """round-4 advice: `torchrun --nproc-per-node 2 bench.py` (no --gpus) used to die on an assert with no JSON line; the launcher's
    WORLD_SIZE is now the statement, and only an EXPLICIT --gpus that disagrees is refused."""
    import subprocess
    from test_bench_contract import strict_line
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, UNC_DIST_BACKEND="gloo", UNC_BENCH_LIB=str(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so"),
               UNC_BENCH_CACHE=str(tmp_path), UNC_BENCH_DETAIL=str(tmp_path / "bench_detail.json"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "example",
                        "--reads", "1"], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    b = strict_line([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2


def test_numa_placement_plan(tmp_path):
    """uncalled_amd/numa.py: a rank's host threads go to the NUMA node of its GPU (sysfs), to an equal share of the allowed
    cores when the platform names no node, and stay put on one GPU -- worked out against a fake sysfs tree."""
    from uncalled_amd import numa
    assert numa.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and numa.parse_cpulist("") == []
    sysfs = tmp_path / "sys"
    for bdf, node in (("0000:c1:00.0", 1), ("0000:05:00.0", -1)):
        d = sysfs / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in ((0, "0-7"), (1, "8-15")):
        d = sysfs / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    allowed = set(range(16))
    cpus, how = numa.plan(3, 8, allowed, "0000:c1:00.0", str(sysfs))
    assert cpus == list(range(8, 16)) and "numa node 1" in how
    cpus, how = numa.plan(3, 8, set(range(4, 12)), "0000:c1:00.0", str(sysfs))       # only the allowed cores of the node
    assert cpus == [8, 9, 10, 11]
    cpus, how = numa.plan(3, 8, allowed, "0000:05:00.0", str(sysfs))                 # node -1: equal shares
    assert cpus == [6, 7] and "share 3 of 8" in how
    cpus, how = numa.plan(0, 1, allowed, None, str(sysfs))                           # one GPU, nothing known: untouched
    assert cpus == sorted(allowed) and how == "left as is"
    out = numa.pin_to_gpu_node(0, 1, str(sysfs))                                     # never raises, also without a GPU
    assert "how" in out


def test_gpu_pci_address_comes_from_the_c_abi(sim_lib):
    """round-3 advice: the address is asked of libuncalled_hip.so (unc_device_pci_address -> hipDeviceGetPCIBusId, lower-cased as
    sysfs spells it), not of torch; the emulator's HIP stub answers 0000:C1:00.0 for device 0."""
    from uncalled_amd import numa
    assert numa.gpu_pci_address(0, lib=sim_lib) == "0000:c1:00.0"
    assert numa.gpu_pci_address(1, lib=sim_lib) == "0000:c2:00.0"
