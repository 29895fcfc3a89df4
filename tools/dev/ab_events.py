"""Dev tool: time k_events alone (unc_detect_events) for several builds of the library on the bench's E. coli reads and compare
the event means / counts with the first build's.

    python tools/dev/ab_events.py <n_reads> <lib.so>[@events_reads_per_wave] ..."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from uncalled_amd import capi
from uncalled_amd.build_index import synthetic_genome
from tools.simulate_reads_torch import simulate_reads_torch
from tools.simulate_reads import CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION

n = int(sys.argv[1])
names, lens, codes = synthetic_genome(1, 4641652, seed=1)
pre = ROOT / "data" / "ecoli_p"
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
raw = sim["signal"].cpu().numpy()
off = sim["offsets"]
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
first = None
for spec in sys.argv[2:]:
    lib, _, rpw = spec.partition("@")
    L = capi.load(lib)
    ix = capi.Index(pre, lib=L)
    m = capi.Mapper(ix, **({"events_reads_per_wave": int(rpw)} if rpw else {}))
    ms = []
    for _ in range(3):
        means, moff, info = m.detect_events(raw, off, cal)
        ms.append(round(m.last_timing()[0], 2))
    same = "ref"
    if first is None:
        first = (means.copy(), moff.copy(), info.copy())
    else:
        ok = np.array_equal(moff, first[1]) and np.array_equal(means.view(np.uint32), first[0].view(np.uint32)) and info.tobytes() == first[2].tobytes()
        same = "IDENTICAL" if ok else "DIFFERENT (expected for the timing experiments)"
    print(Path(lib).name + ("@" + rpw if rpw else ""), "k_events ms:", ms, "events", int(moff[-1]), same, flush=True)
    m.close(); ix.close()
