"""Dev tool: time k_map of several builds of the library on the same 50k-read E. coli batch."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from uncalled_amd import capi
from tools.build_index import build_from_codes, synthetic_genome
from tools.simulate_reads_torch import simulate_reads_torch
from tools.simulate_reads import CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION

n = int(sys.argv[1])
names, lens, codes = synthetic_genome(1, 4641652, seed=1)
pre = Path("/tmp/ub/ecoli_syn"); pre.parent.mkdir(exist_ok=True)
if not Path(str(pre) + ".sa").exists():
    build_from_codes(pre, names, [""], lens, codes)
    from uncalled_amd.index_params import parameterize
    _ix = capi.Index(pre); parameterize(_ix, pre); _ix.close()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
for lib in sys.argv[2:]:
    L = capi.load(lib)
    ix = capi.Index(pre, lib=L)
    m = capi.Mapper(ix)
    t = []
    for i in range(3):
        m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
        t.append(m.last_timing()[1])
    pc = m.last_phase_cycles(); tot = float(sum(pc.values())) or 1.0
    print({k: round(v / tot, 3) for k, v in pc.items()})
    print(lib, "k_map ms:", [round(x, 1) for x in t], flush=True)
    m.close(); ix.close()
