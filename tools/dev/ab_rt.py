"""Dev tool: the realtime workload of bench.py (512 channels x 4000-sample chunks, E. coli thresholds) on several builds of
the library: round latency and the two kernels' times per round.

    python tools/dev/ab_rt.py <rounds> <lib.so> ..."""
import argparse
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench
from uncalled_amd import capi
from uncalled_amd.build_index import synthetic_genome

rounds = int(sys.argv[1])
names, lens, codes = synthetic_genome(1, 4641652, seed=1)
pre = ROOT / "data" / "ecoli_p"
a = argparse.Namespace(channels=512)
for lib in sys.argv[2:]:
    L = capi.load(lib)
    ix = capi.Index(pre, lib=L)
    r = bench.realtime_workload(a, ix, pre, codes, lens, 0, "ecoli", rounds, 3, cpu_budget_s=0.0)
    c = r["config"]
    print(Path(lib).name, {k: round(v, 1) for k, v in c["latency_ms"].items()}, {k: round(v, 2) for k, v in c["kernel_ms"].items()},
          "reads finished", c["reads_finished"], flush=True)
    ix.close()
