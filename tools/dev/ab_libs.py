"""Dev tool: time k_map of several builds of the library on the same batch and check that they agree.

    python tools/dev/ab_libs.py <n_reads>[:workload] <lib.so> [<lib.so> ...]
workload: ecoli (default), chr20, hs400 (400 Mb in 8 contigs: where add_seed is a third of the time) -- bench.py's references.
The first library's hits are the reference the others are compared with (all fields, bit for bit).
AB_NOPROF=1 skips the extra pass with the profiling instantiation (phase shares)."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from uncalled_amd import capi
from uncalled_amd.build_index import build_from_codes, synthetic_genome
from tools.simulate_reads_torch import simulate_reads_torch
from tools.simulate_reads import CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION

n, _, workload = sys.argv[1].partition(":")
n = int(n)
workload = workload or "ecoli"
if workload == "ecoli":
    names, lens, codes = synthetic_genome(1, 4641652, seed=1)
    pre = Path("/tmp/ub/ecoli_syn"); pre.parent.mkdir(exist_ok=True)
    if (ROOT / "data" / "ecoli_p.sa").exists():       # a parameterised index left by an earlier build: travels with the snapshot
        pre = ROOT / "data" / "ecoli_p"
    elif not Path(str(pre) + ".sa").exists():
        build_from_codes(pre, names, [""], lens, codes)
        from uncalled_amd.index_params import parameterize
        _ix = capi.Index(pre); parameterize(_ix, pre); _ix.close()
else:
    import bench       # its index cache and builders (GPU suffix sort + `uncalled index` parameterisation)
    pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
first = None
for spec in sys.argv[2:]:
    # lib.so[@n_slots[@slice_events[@events_reads_per_wave[@n_waves]]]]: n_slots 0 = library default, 1 = one slot per wavefront (no time slicing)
    lib, *rest = spec.split("@")
    kw = {}
    if rest and int(rest[0]) == 1:
        kw = dict(n_slots=256 * 12, n_waves=256 * 12)
    elif rest and int(rest[0]) > 1:
        kw = dict(n_slots=int(rest[0]), n_waves=256 * 12)
    if len(rest) > 1 and int(rest[1]):
        kw["slice_events"] = int(rest[1])
    if len(rest) > 2 and int(rest[2]):
        kw["events_reads_per_wave"] = int(rest[2])
    if len(rest) > 4 and int(rest[4]):          # pairs of scheduler rings (1 = one for the device, 0 = per XCD)
        kw["sched_parts"] = int(rest[4])
    if len(rest) > 3 and int(rest[3]):          # wavefronts in flight (0 = what the kernel's occupancy gives)
        kw["n_waves"] = int(rest[3])
        if kw.get("n_slots") is None and rest and int(rest[0]) == 0:
            kw["n_slots"] = 4 * int(rest[3])
    try:
        L = capi.load(lib)
        ix = capi.Index(pre, lib=L)
        try:
            m = capi.Mapper(ix, **kw)
        except TypeError:
            m = capi.Mapper(ix)
        t, te = [], []
        for i in range(int(os.environ.get("AB_RUNS", 2))):
            hits = m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
            te.append(m.last_timing()[0])
            t.append(m.last_timing()[1])
        busy = m.last_wave_busy() if hasattr(L, "unc_mapper_last_wave_busy") else -1
        if hasattr(L, "unc_mapper_set_profile") and not os.environ.get("AB_NOPROF"):   # phase shares from one extra pass of the counting instantiation
            m.set_profile(True)
            m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal)
            t.append(-m.last_timing()[1])
        pc = m.last_phase_cycles(); tot = float(sum(pc.values())) or 1.0
        same = "ref"
        if first is None:
            first = hits.copy()
        else:
            bad = [f for f in capi.RESULT_FIELDS if not np.array_equal(hits[f], first[f])]
            same = "IDENTICAL" if not bad else "MISMATCH in %s (%d reads)" % (bad, int(sum((hits[f] != first[f]).sum() for f in bad)))
        print({k: round(v / tot, 3) for k, v in pc.items() if v})
        print(spec.split("/")[-1], "k_events ms:", [round(x, 2) for x in te], "k_map ms:", [round(x, 1) for x in t], "wave_busy %.3f" % busy, "slots", m.n_slots if hasattr(m, "n_slots") else "?", "remap", m.last_remap() if hasattr(m, "last_remap") else "?", same, flush=True)
        m.close(); ix.close()
    except Exception as e:   # a bad variant must not hide the others
        print(Path(lib).name, "FAILED:", repr(e)[:300], flush=True)
