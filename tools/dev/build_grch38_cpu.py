"""Dev tool: `grch38_syn` (24 contigs, 3.1 Gbp, seed 3, 30 % masked) built with the chunked builder on the CPU
(torch "cpu"), to exercise uncalled_amd/build_index_big.py past 2^32 symbols without spending GPU minutes.  Takes about
an hour on 8 cores and ~40 GB of RAM; the GPU box does the same in minutes (bench.py --workload grch38)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from uncalled_amd.build_index_big import big_masked_genome, build_from_codes_big  # noqa: E402

out = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/grch38/grch38_syn")
total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3100000000
torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 6)
t0 = time.time()
names, lens, codes, holes, n_ambs = big_masked_genome(24, total, seed=3, name="grch38_syn")
print(f"genome {time.time() - t0:.0f} s", flush=True)
info = build_from_codes_big(out, names, [""] * len(names), lens, codes, holes, n_ambs, device="cpu", verbose=True)
print(info, f"total {time.time() - t0:.0f} s", flush=True)
