"""Dev tool (GPU box): why do GRCh38 launches of ONE library differ by +-10 %?  (round-4 review, item 3)

    python tools/dev/grch38_spread.py [n_reads] [launches]

Maps the bench's own GRCh38 batch (250 000 reads) several times with (A) the pool as rounds 1-4 sized it (60 % of the free HBM,
named explicitly so that the library leaves it alone) and (B) the pool sized by need (default: the first launch runs on the
rule-of-thumb pool, the library then keeps it at twice the high-water mark).  Per launch: k_map ms (HIP events), pool chunks / GB,
free HBM, and the GPU's clocks and power as rocm-smi reports them right after the launch.  Hits of every launch must be identical."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
workload = os.environ.get("SPREAD_WORKLOAD", "grch38")


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        c = d.get("card0", {})
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "junction"))}
        return keep
    except Exception as e:      # noqa
        return {"smi_error": repr(e)[:80]}


pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
ix = capi.Index(pre)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
out = {"workload": workload, "reads": n, "legs": {}}
first = None
legs = [("pool_by_need", {})]
if not os.environ.get("SPREAD_ONLY_NEED"):
    legs.insert(0, ("pool_60pct_of_free", None))
for leg, kw in legs:
    torch.cuda.empty_cache()
    if kw is None:
        free_b, _ = torch.cuda.mem_get_info()
        # what unc_mapper_create's rule gives (slots are allocated first: take them off)
        probe = capi.Mapper(ix)
        rule = probe.geometry()["pool_chunks"]
        probe.close()
        kw = dict(pool_chunks=rule)
    m = capi.Mapper(ix, **kw)
    rows = []
    for i in range(launches):
        t0 = time.perf_counter()
        hits = m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
        wall = time.perf_counter() - t0
        free_b, tot_b = torch.cuda.mem_get_info()
        row = {"launch": i, "k_map_ms": round(m.last_timing()[1], 1), "wall_s": round(wall, 2), "pool": m.pool_usage(), "remap": m.last_remap()[0],
               "free_gb": round(free_b / 1e9, 1), "used_gb": round((tot_b - free_b) / 1e9, 1), "smi": smi()}
        if first is None:
            first = hits.copy()
            row["hits"] = "ref"
        else:
            bad = [f for f in capi.RESULT_FIELDS if not np.array_equal(hits[f], first[f])]
            row["hits"] = "IDENTICAL" if not bad else "MISMATCH " + ",".join(bad)
        rows.append(row)
        print(leg, json.dumps(row), flush=True)
    out["legs"][leg] = rows
    ms = [r["k_map_ms"] for r in rows]
    print(f"== {leg}: k_map ms {ms}  spread (max-min)/median = {100 * (max(ms) - min(ms)) / float(np.median(ms)):.1f} %", flush=True)
    m.close()
print(json.dumps(out))
