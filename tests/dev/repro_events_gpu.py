"""Dev tool (GPU box, no torch): the event detector on dumped reads (tests/dev/dump_reads.py) at every alignment of the read inside its
batch, per library, against the oracle.   python tests/dev/repro_events_gpu.py <dump.npz> <read> <lib.so> [<lib.so> ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from oracle import pyoracle as po
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE

d = np.load(sys.argv[1])
r = int(sys.argv[2])
raw = d[f"raw_{r}"]
sig = po.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
ev, mel, tot = po.detect_events(sig)[:3]
om = ev["mean"].astype(np.float32)
print(f"read {r}: oracle kept {len(ev)} total {tot}; in-batch GPU result of the dump: {d[f'gpu_info_{r}']}", flush=True)
for libp in sys.argv[3:]:
    lib = capi.load(libp)
    ix = capi.Index(ROOT / "tests" / "golden" / "example_index" / "example_ref", lib=lib)
    m = capi.Mapper(ix, n_slots=64)
    for pad in range(0, 9):
        full = np.concatenate((np.full(pad, 500, np.int16), raw))
        off = np.array([0, pad, pad + raw.size], dtype=np.uint64) if pad else np.array([0, raw.size], dtype=np.uint64)
        cal = capi.make_calib(off.size - 1, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
        means, moff, info = m.detect_events(full, off, cal)
        k = off.size - 2
        g = means[int(moff[k]):int(moff[k + 1])]
        first = next((i for i in range(min(len(g), len(om))) if g[i].tobytes() != om[i].tobytes()), None)
        extra = ""
        if first is not None:
            extra = f" gpu[{first}]={g[first]!r} oracle[{first}]={om[first]!r} oracle start {int(ev['start'][first])} len {int(ev['length'][first])}"
        print(f"{Path(libp).name} pad {pad}: kept {info[k]['n_events']} total {info[k]['total_events']} first diff {first}{extra}", flush=True)
    m.close(); ix.close()
