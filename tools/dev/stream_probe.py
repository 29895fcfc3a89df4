"""Dev tool (GPU box): ONE mapper, the same batch launched on each of several streams in turn -- which stream (hardware queue) a persistent
k_map launch is submitted on decides which of its two speeds it runs at (round 6).

    python tools/dev/stream_probe.py [workload = ecoli] [n_reads = 50000] [streams = 8] [rounds = 2]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

workload = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
ix = capi.Index(pre)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
m = capi.Mapper(ix)
streams = [("the mapper's own", None), ("torch's current", torch.cuda.current_stream().cuda_stream)]
keep = []
for i in range(n_streams):
    pr = -1 if i % 2 else 0
    s = torch.cuda.Stream(device=0, priority=pr)
    keep.append(s)
    streams.append((f"new stream {i} (priority {pr})", s.cuda_stream))
for r in range(rounds):
    for name, st in streams:
        m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal, stream=st)
        print(f"round {r}: {name:32s} k_map {m.last_timing()[1]:8.1f} ms  wave_busy {m.last_wave_busy():.3f}", flush=True)
