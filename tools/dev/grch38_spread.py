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


class Sampler:
    """GPU clock / power / temperature from the hwmon files of card 0 every 0.25 s while a launch runs (a thread: no subprocess)."""
    def __init__(self):
        import glob
        self.files = {}
        for pat, key in (("freq1_input", "sclk_hz"), ("freq2_input", "mclk_hz"), ("power1_average", "power_uw"), ("power1_input", "power_uw"),
                         ("temp2_input", "junction_mC"), ("temp1_input", "edge_mC")):
            for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + pat))[:1]:
                self.files.setdefault(key, f)
        self.rows, self.stop = [], False
    def read(self):
        out = {}
        for k, f in self.files.items():
            try:
                out[k] = int(open(f).read().strip())
            except Exception:      # noqa
                pass
        return out
    def run(self):
        import threading
        self.rows, self.stop = [], False
        def loop():
            while not self.stop:
                self.rows.append(self.read()); time.sleep(0.25)
        self.t = threading.Thread(target=loop, daemon=True); self.t.start()
    def end(self):
        self.stop = True; self.t.join()
        out = {"samples": len(self.rows)}
        for k in self.files:
            v = [r[k] for r in self.rows if k in r]
            if v:
                out[k] = {"mean": round(float(np.mean(v)), 1), "min": int(np.min(v)), "max": int(np.max(v))}
        return out


sampler = Sampler()
pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
ix = capi.Index(pre)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
out = {"workload": workload, "reads": n, "legs": {}}
first = None
# SPREAD_LEGS: comma list of leg names; "fixed" = pool named explicitly (rounds 1-4's size), "auto" = sized by need, "fixed2" = a second
# mapper with the named pool created after the first was freed (same process: does a mapper's PLACE in memory decide its speed?)
legs = []
for name in os.environ.get("SPREAD_LEGS", "fixed,auto").split(","):
    legs.append((name, {} if name.startswith("auto") else None))
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
        sampler.run()
        t0 = time.perf_counter()
        hits = m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
        wall = time.perf_counter() - t0
        during = sampler.end()
        free_b, tot_b = torch.cuda.mem_get_info()
        row = {"launch": i, "k_map_ms": round(m.last_timing()[1], 1), "wall_s": round(wall, 2), "wave_busy": round(m.last_wave_busy(), 4), "pool": m.pool_usage(), "remap": m.last_remap()[0], "during": during,
               "free_gb": round(free_b / 1e9, 1), "used_gb": round((tot_b - free_b) / 1e9, 1), "smi": smi() if i == 0 else {}}
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
