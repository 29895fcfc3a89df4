"""`uncalled index` (SURVEY f-2): the .uncl thresholds.  Goldens come from the reference's own Python
IndexParameterizer + C++ self_align executed in place (tests/golden/make_uncl_goldens.py); the example golden is also
byte-identical to the .uncl file the reference ships in example/index/ -- a known answer from the reference's tree."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from uncalled_amd.build_index import build_from_codes, synthetic_genome
from uncalled_amd import capi
from uncalled_amd.index_params import choose_sample_dist, parameterize

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "uncl_goldens.json").read_text())
PRESETS = (("default", dict(tgt_speed=115)), ("speed_60", dict(tgt_speed=60)))


def _digest(lens, full):
    h = hashlib.sha256()
    for row, n in zip(lens, full):
        assert n <= lens.shape[1]
        h.update(np.asarray(row[:n], dtype=np.uint64).tobytes())
        h.update(b"|")
    return h.hexdigest()


def _check(lib, example, tmp_path):
    ix = capi.Index(example["prefix"], lib=lib)
    # self-alignment trajectories equal the reference's (sampling every base)
    lens, full = ix.self_align(example["prefix"], 1, cap=64)
    assert len(full) == 10000 and _digest(lens, full) == GOLD["example"]["self_align_dist1_sha256"]
    p = parameterize(ix, example["prefix"], presets=PRESETS, write=False)
    assert p.text() == GOLD["example"]["uncl"]
    assert p.text().splitlines()[0] + "\n" == GOLD["example"]["bundled"]          # the file the reference ships
    # a 3-contig synthetic reference: contig boundaries + rand() sampling at distance 12
    names, lens_, codes = synthetic_genome(3, 600000, seed=5)
    prefix = tmp_path / "syn600k"
    build_from_codes(prefix, names, [""] * 3, lens_, codes)
    ix2 = capi.Index(prefix, lib=lib)
    assert choose_sample_dist(600000) == 12
    tl, tf = ix2.self_align(prefix, 12, cap=64)
    assert _digest(tl, tf) == GOLD["syn600k_seed5_3contigs"]["self_align_dist12_sha256"]
    p2 = parameterize(ix2, prefix, presets=PRESETS, write=True)
    assert p2.text() == GOLD["syn600k_seed5_3contigs"]["uncl"]
    # and the freshly written .uncl loads
    assert capi.Index(prefix, lib=lib).thresholds()[63] == np.float32(-10.07)


@pytest.mark.lanesim
def test_uncl_matches_reference_lanesim(sim_lib, example, tmp_path):
    _check(sim_lib, example, tmp_path)


@pytest.mark.gpu
def test_uncl_matches_reference_gpu(hip_lib, example, tmp_path):
    _check(hip_lib, example, tmp_path)
