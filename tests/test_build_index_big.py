"""uncalled_amd/build_index_big.py (chunked suffix sort for references past 2^31 symbols) against the bundled `bwa index` files
and against uncalled_amd/build_index.py on repeat-rich synthetic genomes, with chunks and pieces small enough to exercise every
boundary (torch on the CPU here; the same code runs on the GPU)."""
import filecmp
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uncalled_amd import build_index as small   # noqa: E402
from uncalled_amd import build_index_big as big   # noqa: E402

EX = ROOT / "tests" / "golden" / "example_index"
SUFS = (".pac", ".ann", ".amb", ".bwt", ".sa")


def test_example_index_is_reproduced(tmp_path):
    names, annos, seqs = small.read_fasta(EX / "example_ref.fa")
    codes, holes, n_ambs = small.encode_contigs(seqs)
    big.build_from_codes_big(tmp_path / "x", names, annos, [len(s) for s in seqs], codes, holes, n_ambs, uncl_text=None, device="cpu",
                             chunk=3000, piece=4096)
    for suf in SUFS:
        assert filecmp.cmp(tmp_path / ("x" + suf), EX / ("example_ref" + suf), shallow=False), suf


@pytest.mark.parametrize("seed,n_contigs,total,chunk,piece", [(5, 3, 40000, 7000, 8192), (6, 1, 30011, 100000, 1 << 20), (7, 2, 25000, 2000, 5000)])
def test_matches_small_builder_on_repeats(tmp_path, seed, n_contigs, total, chunk, piece):
    rng = np.random.default_rng(seed)
    names, lens, codes = small.synthetic_genome(n_contigs, total, seed)
    codes = codes.copy()
    # long exact repeats, a tandem repeat and a homopolymer run: ties far deeper than one 21-symbol key
    codes[9000:12000] = codes[2000:5000]
    unit = rng.integers(0, 4, 37).astype(np.uint8)
    codes[15000:15000 + 37 * 40] = np.tile(unit, 40)
    codes[20000:20900] = 0
    small.build_from_codes(tmp_path / "a", names, [""] * len(names), lens, codes, uncl_text=None)
    big.build_from_codes_big(tmp_path / "b", names, [""] * len(names), lens, codes, uncl_text=None, device="cpu", chunk=chunk, piece=piece)
    for suf in SUFS:
        assert filecmp.cmp(tmp_path / ("a" + suf), tmp_path / ("b" + suf), shallow=False), suf
