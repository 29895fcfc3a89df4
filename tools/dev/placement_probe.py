"""Dev tool (GPU box): does WHERE a mapper's memory lies decide which of its two speeds it runs at?  (Since round 4: the same library maps the
same batch 2 - 8 % faster or slower from one mapper instance to the next, steadily for the life of the instance.)

    python tools/dev/placement_probe.py [workload = ecoli] [n_reads = 50000] [instances = 8]

Creates mapper instances one after the other (each freed before the next, with an odd-sized spacer allocation in between on odd rounds),
maps the bench batch twice on each and prints k_map ms beside the device addresses of the slots, the pool, the event means."""
import ctypes as C
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from uncalled_amd import capi
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE
from tools.simulate_reads_torch import simulate_reads_torch

workload = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
inst = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pre, codes, lens = bench.ensure_index(Path("/tmp/uncalled_amd_bench"), 0, lambda: None, workload, "cuda:0")
ix = capi.Index(pre)
torch.cuda.empty_cache()
sim = simulate_reads_torch(codes, lens, n, seed=42, device="cuda:0")
del codes
cal = capi.make_calib(n, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION)
L = ix.L
L.unc_mapper_device_addresses.argtypes = [C.c_void_p, C.c_void_p]
import os
FIXED = torch.cuda.Stream(device=0) if os.environ.get("PROBE_STREAM") else None


class _Dev:
    """a device allocation of the library seen as a torch tensor (uint64 words), through __cuda_array_interface__"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes) // 8,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


_idx = None


def gather_ms(ptr, nbytes, n_idx=1 << 26):
    """random 8-byte reads over an allocation: what address translation over it costs (ms for 2^26 reads, the best of three)"""
    global _idx
    t = torch.as_tensor(_Dev(ptr, nbytes), device="cuda:0")
    if _idx is None or int(_idx.max()) >= t.numel():
        _idx = torch.randint(0, t.numel(), (n_idx,), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s = t[_idx].sum()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def run(tag, m):
    t = []
    for _ in range(2):
        # PROBE_STREAM=1: every instance launches on ONE stream of the caller's instead of the stream each mapper creates for itself
        m.map_batch_device(sim["signal"].data_ptr(), sim["offsets"], cal, stream=(FIXED.cuda_stream if FIXED is not None else None))
        t.append(round(m.last_timing()[1], 1))
    a = np.zeros(6, dtype=np.uint64)
    L.unc_mapper_device_addresses(m.h, a.ctypes.data)
    L.unc_calib_chase.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    ch = (C.c_float * 2)()
    L.unc_calib_chase(0, int(a[0]), int(a[1]) * m.geometry()["n_slots"], 4096, 2000, C.byref(ch, 0))
    L.unc_calib_chase(0, int(a[2]), int(a[3]), 4096, 2000, C.byref(ch, 4))
    g_slots = gather_ms(int(a[0]), int(a[1]) * m.geometry()["n_slots"])
    g_pool = gather_ms(int(a[2]), int(a[3]))
    free_b, _ = torch.cuda.mem_get_info(0)
    print(f"{tag}: k_map ms {t}  dependent 16-byte loads (4 096 wavefronts x 2 000 steps): slots {ch[0]:.2f} ms, pool {ch[1]:.2f} ms;  random 8-byte reads, 2^26 of them: over the slots {g_slots:.2f} ms, over the pool {g_pool:.2f} ms   wave_busy {m.last_wave_busy():.3f}  slots at {int(a[0]):#x} pool at {int(a[2]):#x} means at {int(a[4]):#x}  free {free_b / 1e9:.1f} GB", flush=True)


# plan: comma-separated steps (argv[4]); plain = create, map twice, free; twice = create two, free the first, map on the second;
# dummyN = hold an N-GB torch allocation while the mapper is created, free it, then map; keep = create and map, never free
plan = (sys.argv[4] if len(sys.argv) > 4 else "plain,plain,plain,plain,twice,twice,twice,dummy60,dummy60,dummy60,plain,plain").split(",")
kept = []
for i, step in enumerate(plan):
    if step == "plain":
        m = capi.Mapper(ix); run(f"{i} plain", m); m.close()
    elif step == "twice":
        m1 = capi.Mapper(ix); m2 = capi.Mapper(ix); m1.close(); run(f"{i} twice (the second of two, the first freed)", m2); m2.close()
    elif step.startswith("dummy"):
        d = torch.empty(int(step[5:]) << 30, dtype=torch.uint8, device="cuda:0")
        m = capi.Mapper(ix)
        del d
        torch.cuda.empty_cache()
        run(f"{i} {step} (created beside a dummy allocation, since freed)", m); m.close()
    elif step == "keep":
        m = capi.Mapper(ix); run(f"{i} keep", m); kept.append(m)
