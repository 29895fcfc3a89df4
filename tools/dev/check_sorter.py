"""Dev tool (GPU box): the chr20-sized index (129 M symbols) built with the two sorters of uncalled_amd/build_index.py -- k_sort.hip's radix
sort (unc_sort_pairs_u64) and torch.sort -- must come out byte-identical; prints the build times.   python tools/dev/check_sorter.py [workload]"""
import hashlib
import sys
import tempfile
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402
from uncalled_amd.build_index import build_from_codes, masked_synthetic_genome, synthetic_genome  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "chr20"
if wl == "grch38":       # the chunked builder (6.2 G symbols, 2^28-key chunks): once per sorter
    from uncalled_amd.build_index_big import big_masked_genome, build_from_codes_big
    names, lens, codes, holes, n_ambs = big_masked_genome(24, 3100000000, seed=3, name="grch38_syn")
    out = {}
    with tempfile.TemporaryDirectory(prefix="unc_sorter_", dir="/tmp") as d:
        for sorter in ("hip", "torch"):
            pre = Path(d) / sorter
            t0 = time.time()
            build_from_codes_big(pre, names, [""] * len(names), lens, codes, holes, n_ambs, uncl_text=None, device="cuda", sorter=sorter)
            torch.cuda.synchronize()
            secs = time.time() - t0
            dig = {}
            for suf in (".bwt", ".sa"):
                h = hashlib.sha256()
                with open(str(pre) + suf, "rb") as f:
                    for blk in iter(lambda: f.read(1 << 26), b""):
                        h.update(blk)
                dig[suf] = h.hexdigest()[:16]
            print(sorter, f"{secs:.1f} s", dig, flush=True)
            out[sorter] = dig
            for suf in (".bwt", ".sa", ".pac", ".ann", ".amb"):
                Path(str(pre) + suf).unlink(missing_ok=True)
    print("IDENTICAL" if out["hip"] == out["torch"] else "MISMATCH")
    sys.exit(0)
if wl == "chr20":
    names, lens, codes, holes, n_ambs = masked_synthetic_genome(1, 64444167, seed=2, name="chr20_syn")
else:
    names, lens, codes = synthetic_genome(1, 4641652, seed=1)
    holes, n_ambs = (), None
out = {}
with tempfile.TemporaryDirectory(prefix="unc_sorter_") as d:
    for sorter in ("hip", "torch", "hip"):
        pre = Path(d) / sorter
        t0 = time.time()
        build_from_codes(pre, names, [""] * len(names), lens, codes, holes, n_ambs, uncl_text=None, sa_device="cuda", sorter=sorter)
        torch.cuda.synchronize()
        secs = time.time() - t0
        dig = {suf: hashlib.sha256(Path(str(pre) + suf).read_bytes()).hexdigest()[:16] for suf in (".bwt", ".sa", ".pac")}
        print(sorter, f"{secs:.1f} s", dig, flush=True)
        out.setdefault(sorter, dig)
print("IDENTICAL" if out["hip"] == out["torch"] else "MISMATCH")
