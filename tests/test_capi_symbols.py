"""CPU: the gfx950 library loads and exports every entry point include/uncalled_hip.h declares (no compute
calls without a GPU), and the package refuses to work without it."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    txt = (ROOT / "include" / "uncalled_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(unc_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("unc_index_load", "unc_mapper_create", "unc_map_batch", "unc_map_batch_begin", "unc_map_batch_end", "unc_detect_events", "unc_params_default"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    lib = g.build_hip()          # hipcc cross-compiles for gfx950 without a GPU
    L = ctypes.CDLL(str(lib))
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    L.unc_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.unc_version()


def test_params_default_match_reference_defaults():
    from uncalled_amd import capi
    p = capi.default_params()
    # mapper.cpp:29-40, event_detector.cpp:17-26, seed_tracker.cpp:28-32, read_buffer.cpp:26-32
    assert (p.seed_len, p.max_rep_copy, p.max_paths, p.max_consec_stay, p.max_events) == (22, 50, 10000, 8, 30000)
    assert (p.window_length1, p.window_length2, p.min_map_len) == (3, 6, 25)
    assert abs(p.min_seed_prob + 3.75) < 1e-7 and abs(p.threshold1 - 1.4) < 1e-6 and abs(p.min_top_conf - 1.85) < 1e-6
    assert (p.bp_per_sec, p.sample_rate) == (450.0, 4000.0)


def test_missing_library_fails_loudly(tmp_path):
    from uncalled_amd import capi
    with pytest.raises(capi.UncalledHipError):
        capi.load(tmp_path / "libuncalled_hip.so")


def test_product_package_does_not_reference_the_oracle():
    """The product path may not import, link or execute anything under oracle/ (or the lanesim emulator)."""
    for f in list((ROOT / "uncalled_amd").rglob("*.py")) + list((ROOT / "uncalled_amd" / "csrc").rglob("*")):
        if f.suffix in (".so", ".o") or f.is_dir():
            continue
        txt = f.read_text(errors="ignore")
        assert "oracle" not in txt.replace("no CPU", "") or f.name == "r94_model_table.h", f
        assert "unc_o_" not in txt and "pyoracle" not in txt and "pyref" not in txt, f


def test_only_the_allowed_places_use_the_oracle():
    """Besides the product package (above): `tools/` (data tooling, dev scripts) does not import the oracle either -- scripts
    that use it as a checker live under tests/.  bench.py touches it only in its cpu_baseline leg, __graft_entry__ only in
    build() (compiling the checker) and smoke()."""
    import re
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    for f in (ROOT / "tools").rglob("*.py"):
        assert not pat.search(f.read_text()), f
    bench = (ROOT / "bench.py").read_text()
    uses = [m.start() for m in pat.finditer(bench)]
    leg = bench.index("def cpu_baseline(")
    end = bench.index("\ndef ", bench.index("def cpu_baseline_subprocess("))
    assert uses and all(leg < u < end for u in uses), "bench.py imports the oracle outside its cpu_baseline leg"
