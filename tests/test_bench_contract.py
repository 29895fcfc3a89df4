"""The one JSON line `bench.py` prints (driver contract + tier additions), checked on the committed line of the round
(profiles/r01_bench_default_final.json, produced on the GPU box) and on bench.py's own helpers."""
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def test_committed_bench_line_has_every_contract_field():
    b = json.loads((ROOT / "profiles" / "r01_bench_default_final.json").read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["metric"] == "reads_mapped_per_sec" and b["unit"] == "reads/s" and b["higher_is_better"] is True
    assert b["scaling"] == "weak" and b["vs_baseline"] is None and b["data"] == "synthetic" and b["n_gpus"] == 1
    assert "workload" in b["config"] and "model" not in b["config"]
    # value is whole-job throughput over the timed steps
    assert abs(b["value"] - b["config"]["reads_per_gpu_per_step"] * b["n_gpus"] / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-6
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"]
    c = b["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["paf_mismatches_vs_gpu"] == 0
    # the rocprofv3 summary of the same command agrees with the HIP-event launch time
    stats = (ROOT / "profiles" / "r01_rocprofv3_kernel_stats_final.csv").read_text().splitlines()
    row = next(l for l in stats if "k_map<false>" in l)
    avg_ms = float(row.split(",")[-5]) * 1e-6       # AverageNs
    assert abs(avg_ms - r["launch_ms"]) / r["launch_ms"] < 0.02


def test_algorithmic_bytes_formula():
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    hits = np.zeros(2, dtype=[("n_events", "<u4"), ("event_i", "<u4"), ("n_nbr", "<u8"), ("n_lf", "<u8"), ("n_sa", "<u8"), ("mapped", "<i4")])
    hits["n_events"] = [10, 20]; hits["event_i"] = [5, 20]; hits["n_nbr"] = [100, 7]; hits["n_lf"] = [3, 0]; hits["n_sa"] = [1, 0]
    hits["mapped"] = [1, 0]
    off = np.array([0, 1000, 3000], dtype=np.uint64)
    ev, mp = mod.algorithmic_bytes(hits, off)
    # k_events 2 S + 4 E_kept + 24; k_map 4 E_popped + 128 N_nbr + 64 N_lf + 8 N_sa + 64
    assert ev == (2 * 1000 + 4 * 10 + 24) + (2 * 2000 + 4 * 20 + 24)
    assert mp == (4 * 5 + 128 * 100 + 64 * 3 + 8 * 1 + 64) + (4 * 20 + 128 * 7 + 0 + 0 + 64)
