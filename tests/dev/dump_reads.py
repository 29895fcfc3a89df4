"""Dev tool (GPU box): regenerate the bench batch of a workload (no index needed) and dump chosen reads for local debugging:
raw int16 signal, its offset in the batch modulo 8 (k_events takes aligned 16-byte loads), and the event detector's result on the
GPU for the read IN ITS BATCH POSITION next to the oracle's (events kept / total, first differing kept event).

    python tests/dev/dump_reads.py <workload> <out.npz> <read> [<read> ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from oracle import pyoracle as po
from uncalled_amd import capi
from uncalled_amd.build_index import masked_synthetic_genome, synthetic_genome
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

workload, out = sys.argv[1], sys.argv[2]
reads = [int(x) for x in sys.argv[3:]]
n = {"ecoli": 50000, "chr20": 200000, "grch38": 250000}[workload]
if workload == "chr20":
    names, lens, codes, holes, n_ambs = masked_synthetic_genome(1, 64444167, seed=2, name="chr20_syn")
elif workload == "grch38":
    from uncalled_amd.build_index_big import big_masked_genome
    names, lens, codes, holes, n_ambs = big_masked_genome(24, 3100000000, seed=3, name="grch38_syn")
else:
    names, lens, codes = synthetic_genome(1, 4641652, seed=1)
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
off = sim["offsets"].astype(np.int64)
ix = capi.Index(ROOT / "tests" / "golden" / "example_index" / "example_ref")     # k_events does not look at the index
m = capi.Mapper(ix, n_slots=64)
save = {}
for r in reads:
    lo = max(0, r - 2)
    a = int(off[lo]) // 8 * 8                       # keep the alignment the read has in the batch
    raw = sim["signal"][a:int(off[r + 1])].cpu().numpy()
    o = np.array([0] + [int(off[j]) - a for j in range(lo, r + 2)], dtype=np.uint64) if int(off[lo]) > a else np.array([int(off[j]) - a for j in range(lo, r + 2)], dtype=np.uint64)
    cal = capi.make_calib(o.size - 1, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    means, moff, info = m.detect_events(raw, o, cal)
    k = o.size - 2                                   # the target is the last read of the little batch
    g = means[int(moff[k]):int(moff[k + 1])]
    sig = po.calibrate(raw[int(o[k]):int(o[k + 1])], CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
    ev, mel, tot = po.detect_events(sig)[:3]
    om = np.array([e for e in ev["mean"]], dtype=np.float32)
    first = next((i for i in range(min(len(g), len(om))) if g[i].tobytes() != om[i].tobytes()), None)
    print(f"read {r}: samples {int(off[r + 1] - off[r])}, offset % 8 = {int(off[r]) % 8}; GPU kept {info[k]['n_events']} total {info[k]['total_events']} len_sum {info[k]['len_sum']}; "
          f"oracle events {len(ev)} total {tot} mel {mel}; first differing kept event {first}", flush=True)
    save[f"raw_{r}"] = raw[int(o[k]):int(o[k + 1])]
    save[f"offmod_{r}"] = int(off[r]) % 8
    save[f"gpu_means_{r}"] = g
    save[f"gpu_info_{r}"] = info[k:k + 1]
np.savez_compressed(out, **save)
print("saved", out)
