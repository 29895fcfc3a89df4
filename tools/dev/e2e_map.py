"""Dev tool (GPU box): end-to-end `python -m uncalled_amd map` throughput from multi-fast5 files on disk (HDF5 read ->
page-locked staging -> PCIe -> kernels -> PAF text), next to the HBM-resident number of bench.py.

    python tools/dev/e2e_map.py <n_reads>"""
import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401

import bench  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE  # noqa: E402
from tools.simulate_reads_torch import simulate_reads_torch  # noqa: E402
from uncalled_amd import _uncalled_amd as unc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cache = Path("/tmp/uncalled_amd_bench")
prefix, codes, lens = bench.ensure_index(cache, 0, lambda: None, "ecoli", "cuda:0")
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
raw = sim["signal"].cpu().numpy()
off = sim["offsets"]
d = Path("/tmp/unc_e2e")
d.mkdir(exist_ok=True)
per_file = 4000
t0 = time.time()
files = []
for f0 in range(0, n, per_file):
    reads = [dict(id="r%07d" % i, channel=1 + i % 512, number=i, start=0, range=CAL_RANGE, offset=CAL_OFFSET, digitisation=CAL_DIGITISATION,
                  signal=raw[int(off[i]):int(off[i + 1])]) for i in range(f0, min(n, f0 + per_file))]
    fn = d / ("batch_%03d.fast5" % (f0 // per_file))
    unc.write_fast5(str(fn), reads, True, 4000.0)
    files.append(fn)
t_write = time.time() - t0
del sim
torch.cuda.empty_cache()
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "uncalled_amd", "map", str(prefix), str(d)], cwd=str(ROOT), capture_output=True, text=True)
dt = time.time() - t0
lines = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
mapped = sum(1 for l in lines if l.split("\t")[2] != "*")
print(json.dumps({"workload": "python -m uncalled_amd map <ecoli_syn> <dir of multi-fast5 files>", "reads": n, "fast5_files": len(files),
                  "fast5_bytes": sum(f.stat().st_size for f in files), "wall_s_incl_process_start_and_index_load": dt,
                  "reads_per_sec_end_to_end": len(lines) / dt, "paf_lines": len(lines), "mapped": mapped, "rc": r.returncode,
                  "fast5_write_s": t_write, "stderr_tail": r.stderr[-300:]}))
