"""-m gpu: the DEVICE paths of the BWA-format index builders (replacement for bwa_idx_build, bwa_index.hpp:92-101).

Every scale test and every bench block maps against an index that uncalled_amd/build_index.py (`sa_device="cuda"`) or
uncalled_amd/build_index_big.py (`device="cuda"`) built on the GPU; the CPU suite pins the same code on torch-CPU / numpy only.  Here
the device builds of a 2 Mb repeat-rich, masked reference are compared byte for byte -- all five files -- with the numpy builder
(itself byte-identical to `bwa index` on the bundled example, tests/test_build_index_big.py, tests/test_oracle.py), the suffix
array of a device-built 20 kb index with a naive suffix array, and the C ABI loads the device-built index and answers FM queries
as the oracle does on the CPU-built one."""
import filecmp
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from uncalled_amd import build_index as small   # noqa: E402
from uncalled_amd import build_index_big as big   # noqa: E402

pytestmark = pytest.mark.gpu
SUFS = (".pac", ".ann", ".amb", ".bwt", ".sa")


def repeat_rich_masked(total, seed, n_contigs=3):
    """i.i.d. contigs with 30 % N-runs (.amb holes) + long exact repeats, an inverted repeat (a reverse-complement copy: ties between
    the two strands of the text), a tandem repeat and a homopolymer run -- ties far deeper than one 21-symbol key"""
    names, lens, codes, holes, n_ambs = small.masked_synthetic_genome(n_contigs, total, seed, mean_run=2000, name="rep")
    codes = codes.copy()
    rng = np.random.default_rng(seed)
    q = total // 8
    rep, inv, tan, hom = min(30000, q // 2), min(12000, q // 2), min(300, q // 2 // 53), min(5000, q // 4)
    codes[q:q + rep] = codes[3 * q:3 * q + rep]                          # exact repeat (30 kb on the 2 Mb reference)
    codes[5 * q:5 * q + inv] = 3 - codes[2 * q:2 * q + inv][::-1]        # inverted repeat
    unit = rng.integers(0, 4, 53).astype(np.uint8)
    codes[6 * q:6 * q + 53 * tan] = np.tile(unit, tan)                   # tandem repeat
    codes[7 * q:7 * q + hom] = 0                                         # poly-A
    return names, lens, codes, holes, n_ambs


def test_device_builders_equal_the_numpy_builder_byte_for_byte(tmp_path):
    import torch
    assert torch.cuda.is_available()
    names, lens, codes, holes, n_ambs = repeat_rich_masked(2_000_003, seed=11)
    annos = [""] * len(names)
    small.build_from_codes(tmp_path / "cpu", names, annos, lens, codes, holes, n_ambs, uncl_text=None)                       # numpy
    small.build_from_codes(tmp_path / "dev", names, annos, lens, codes, holes, n_ambs, uncl_text=None, sa_device="cuda")     # unc_build_suffix_array (default: no torch)
    small.build_from_codes(tmp_path / "devh", names, annos, lens, codes, holes, n_ambs, uncl_text=None, sa_device="cuda", sorter="hip")     # torch between k_sort.hip's sorts
    small.build_from_codes(tmp_path / "devt", names, annos, lens, codes, holes, n_ambs, uncl_text=None, sa_device="cuda", sorter="torch")
    big.build_from_codes_big(tmp_path / "bigh", names, annos, lens, codes, holes, n_ambs, uncl_text=None, device="cuda", chunk=1 << 18, piece=1 << 17,
                             sorter="hip")                                                                                      # the big builder on k_sort.hip
    # chunk / piece far below the defaults: dozens of chunk and piece boundaries on 4 M symbols, as 6.2 G symbols have with the defaults
    big.build_from_codes_big(tmp_path / "big", names, annos, lens, codes, holes, n_ambs, uncl_text=None, device="cuda", chunk=1 << 18, piece=1 << 17)
    big.build_from_codes_big(tmp_path / "bigd", names, annos, lens, codes, holes, n_ambs, uncl_text=None, device="cuda")     # defaults: one chunk
    for other in ("dev", "devh", "devt", "big", "bigh", "bigd"):
        for suf in SUFS:
            assert filecmp.cmp(tmp_path / ("cpu" + suf), tmp_path / (other + suf), shallow=False), (other, suf)


def test_index_build_on_the_gpu_without_torch(tmp_path):
    """`uncalled index` on the GPU in a process that CANNOT import torch (UNCALLED_AMD_NO_TORCH=1 and a torch.py on the path that raises):
    the suffix sort is unc_build_suffix_array -- the radix sort and the steps between the sorts as HIP kernels behind the C ABI
    (k_sort.hip) -- and the five files equal the numpy builder's byte for byte (round-5 review: only the sort was HIP, everything
    between the sorts torch tensor ops)."""
    import os
    import subprocess
    names, lens, codes, holes, n_ambs = repeat_rich_masked(2_000_003, seed=11)
    annos = [""] * len(names)
    small.build_from_codes(tmp_path / "cpu", names, annos, lens, codes, holes, n_ambs, uncl_text=None)                       # numpy
    block = tmp_path / "no_torch"
    block.mkdir()
    (block / "torch.py").write_text("raise ImportError('torch is blocked in this process')\n")
    script = tmp_path / "build.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from uncalled_amd import build_index as small\n"
        "from tests.test_gpu_index_build import repeat_rich_masked\n"
        "names, lens, codes, holes, n_ambs = repeat_rich_masked(2_000_003, seed=11)\n"
        "t0 = time.time()\n"
        f"small.build_from_codes({str(tmp_path / 'nt')!r}, names, [''] * len(names), lens, codes, holes, n_ambs, uncl_text=None, sa_device='cuda')\n"
        "assert 'torch' not in sys.modules, 'torch was imported'\n"
        "print('built without torch in %.2f s' % (time.time() - t0))\n")
    env = dict(os.environ, UNCALLED_AMD_NO_TORCH="1", PYTHONPATH=str(block) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "built without torch" in r.stdout, (r.stdout[-400:], r.stderr[-800:])
    assert "the torch construction instead" not in r.stderr, r.stderr[-400:]
    for suf in SUFS:
        assert filecmp.cmp(tmp_path / ("cpu" + suf), tmp_path / ("nt" + suf), shallow=False), suf
    # ... and the command itself, `uncalled index` (scripts/uncalled:38-78), in the same torch-less environment on the bundled reference:
    # files equal to the bundled `bwa index` output, .uncl equal to the one the reference shipped
    import shutil
    ex = ROOT / "tests" / "golden" / "example_index"
    fa = tmp_path / "example_ref.fa"
    shutil.copyfile(ex / "example_ref.fa", fa)
    r = subprocess.run([sys.executable, "-m", "uncalled_amd", "index", str(fa)], cwd=str(ROOT), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "the torch construction instead" not in r.stderr, r.stderr[-2000:]
    for suf in SUFS:
        assert (tmp_path / ("example_ref.fa" + suf)).read_bytes() == (ex / ("example_ref" + suf)).read_bytes(), suf
    assert (tmp_path / "example_ref.fa.uncl").read_text() == (ex / "example_ref.uncl").read_text()


def test_radix_sort_against_numpy(hip_lib):
    """unc_sort_pairs_u64 (k_sort.hip), the sort under the device builders: stable, any number of key bits, values carried or generated
    (argsort), ties by the thousand, sizes around the 2 048-pair tiles and a few million pairs -- against numpy's stable argsort."""
    from tests.parity_cases import case_radix_sort
    case_radix_sort(hip_lib, big=True)


def test_device_built_suffix_array_against_naive(tmp_path, hip_lib, oracle_lib):
    """20 kb slice: SA of every row of the device-built index (through the C ABI on the GPU: unc_fm_sa) == naive suffix array of
    fwd + revcomp; the oracle on the same files agrees row by row (as tests/test_oracle.py does for the bundled `bwa index` files)."""
    from uncalled_amd import capi
    names, lens, codes, holes, n_ambs = repeat_rich_masked(20_000, seed=12, n_contigs=1)
    small.build_from_codes(tmp_path / "s", names, [""], lens, codes, holes, n_ambs, sa_device="cuda")
    big.build_from_codes_big(tmp_path / "b", names, [""], lens, codes, holes, n_ambs, device="cuda", chunk=3000, piece=4096)
    for suf in SUFS:
        assert filecmp.cmp(tmp_path / ("s" + suf), tmp_path / ("b" + suf), shallow=False), suf
    n = codes.size
    t = np.concatenate((codes, 3 - codes[::-1]))
    s = bytes(t + 1)
    naive = sorted(range(2 * n + 1), key=lambda i: s[i:])
    oix = oracle_lib.Index(tmp_path / "s")
    assert oix.size == 2 * n
    assert [oix.sa(k) for k in range(1, 2 * n + 1)] == naive[1:]
    ix = capi.Index(tmp_path / "s", lib=hip_lib)
    rows = np.arange(1, 2 * n + 1, dtype=np.uint64)
    assert np.array_equal(ix.sa(rows), np.array(naive[1:], dtype=np.uint64))
    assert np.array_equal(ix.kmer_ranges(), oix.kmer_ranges())
