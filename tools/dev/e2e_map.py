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
# ---- how fast the files can be read at all (Fast5Reader alone, one thread, page cache warm from the write): the loader's ceiling
t0 = time.time()
conf = unc.Conf()
rd = unc.Fast5Reader(conf)
for f in files:
    rd.add_fast5(str(f))
n_read = 0
while True:
    rd.fill_buffer()
    if rd.buffer_size() == 0:
        break
    while rd.buffer_size():
        rd.pop_read()
        n_read += 1
t_reader = time.time() - t0

# ---- the CLI: PAF lines are time-stamped as they arrive, so that start-up (interpreter, index load, dense SA, first batch) and the
# steady state can be told apart
import select
t0 = time.time()
extra = sys.argv[2:]          # further arguments of `uncalled_amd map`, e.g. -t 2 (independent read order instead of the default -t 1 order)
p = subprocess.Popen([sys.executable, "-u", "-m", "uncalled_amd", "map", str(prefix), str(d)] + extra, cwd=str(ROOT), stdout=subprocess.PIPE,
                     stderr=subprocess.PIPE, text=True, bufsize=1)
stamps, mapped, n_lines = [], 0, 0
for line in p.stdout:
    if not line or line.startswith("#"):
        continue
    n_lines += 1
    if line.split("\t")[2] != "*":
        mapped += 1
    if n_lines % 1000 == 1:
        stamps.append((time.time() - t0, n_lines))
err = p.stderr.read()
p.wait()
dt = time.time() - t0
stamps.append((dt, n_lines))
t_first = stamps[0][0]
# PAF lines arrive batch by batch (a burst per unc_map_batch call).  Steady state = everything after the FIRST burst (whose
# arrival time holds the interpreter start, the index load and the first batch's load): lines after it / time after it.
bursts, cur = [], [stamps[0]]
for a, b in zip(stamps, stamps[1:]):
    if b[0] - a[0] > 0.2:
        bursts.append(cur[-1]); cur = []
    cur.append(b)
bursts.append(cur[-1] if cur else stamps[-1])
steady = None
if len(bursts) > 1 and bursts[-1][0] > bursts[0][0]:
    steady = (bursts[-1][1] - bursts[0][1]) / (bursts[-1][0] - bursts[0][0])
print(json.dumps({"workload": "python -m uncalled_amd map <ecoli_syn> <dir of multi-fast5 files> " + " ".join(extra), "reads": n, "fast5_files": len(files),
                  "fast5_bytes": sum(f.stat().st_size for f in files), "wall_s_incl_process_start_and_index_load": dt,
                  "reads_per_sec_end_to_end": n_lines / dt, "first_paf_line_after_s": t_first,
                  "reads_per_sec_steady_state": steady,
                  "steady_state_note": "PAF lines after the first batch's burst / time after it (bursts = runs of lines less than 0.2 s apart)",
                  "bursts_end_s_and_lines": [(round(t, 2), k) for t, k in bursts],
                  "fast5_reader_alone_reads_per_sec": n_read / t_reader if t_reader > 0 else None,
                  "fast5_reader_note": "Fast5Reader.pop_read over the same files on one thread, nothing else running: the ceiling of MapPool's loader thread",
                  "paf_lines": n_lines, "mapped": mapped, "rc": p.returncode,
                  "fast5_write_s": t_write, "stderr_tail": err[-300:]}))
