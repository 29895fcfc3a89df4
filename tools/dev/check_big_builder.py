"""Dev tool (GPU box): uncalled_amd/build_index_big.py against uncalled_amd/build_index.py on a chr20-sized masked reference, both on the GPU."""
import filecmp, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from uncalled_amd import build_index as small, build_index_big as big
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64444167
names, lens, codes, holes, n_ambs = small.masked_synthetic_genome(3, n, seed=2, name="x")
d = tempfile.mkdtemp()
t0 = time.time(); small.build_from_codes(d + "/a", names, [""] * 3, lens, codes, holes, n_ambs, uncl_text=None, sa_device="cuda"); t1 = time.time()
big.build_from_codes_big(d + "/b", names, [""] * 3, lens, codes, holes, n_ambs, uncl_text=None, device="cuda", chunk=1 << 25, piece=1 << 26); t2 = time.time()
print("small %.1f s, big %.1f s" % (t1 - t0, t2 - t1), {s: filecmp.cmp(d + "/a" + s, d + "/b" + s, shallow=False) for s in (".pac", ".ann", ".amb", ".bwt", ".sa")})
