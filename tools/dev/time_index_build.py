"""Suffix-array build time of the chr20-sized reference (GPU box): unc_build_suffix_array (the C ABI, no torch between the sorts) beside the
torch construction around the same radix sort (round 5's default) -- same text, arrays compared.

    python tools/dev/time_index_build.py [total bases = 64444167]
"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from uncalled_amd import build_index as small
from uncalled_amd import capi

total = int(sys.argv[1]) if len(sys.argv) > 1 else 64444167
names, lens, codes, holes, n_ambs = small.masked_synthetic_genome(1, total, seed=2, name="chr20_syn")
t = np.concatenate((codes, (3 - codes)[::-1])).astype(np.uint8)     # the text bwa sorts: forward + reverse complement (bwa_index.hpp:92-101)
L = capi.load()
torch.cuda.synchronize()
res = {}
for name, fn in (("unc_build_suffix_array", lambda: capi.build_suffix_array(t, 0, L)),
                 ("torch around k_sort.hip", lambda: small.suffix_array_torch(t, "cuda:0", "hip")),
                 ("unc_build_suffix_array again", lambda: capi.build_suffix_array(t, 0, L))):
    t0 = time.time()
    sa = fn()
    torch.cuda.synchronize()
    res[name] = (time.time() - t0, np.asarray(sa))
    print(f"{name:32s} {res[name][0]:6.2f} s  ({t.size} symbols)", flush=True)
a = res["unc_build_suffix_array"][1]
b = res["torch around k_sort.hip"][1]
print("arrays equal:", bool(np.array_equal(a, b)), a.size, b.size)
