"""Every data file the package opens at run time must be tracked in git (a fresh clone has to work) and must be
reproducible by a committed generator script."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _git(*args):
    return subprocess.run(["git", "-C", str(ROOT), *args], capture_output=True, text=True)


@pytest.fixture(scope="module")
def tracked():
    r = _git("ls-files")
    if r.returncode != 0:
        pytest.skip("not a git checkout (gpurun snapshot)")
    return set(r.stdout.split("\n"))


def test_package_data_is_tracked(tracked):
    from uncalled_amd import index_params
    data_dir = ROOT / "uncalled_amd" / "data"
    files = [p for p in data_dir.rglob("*") if p.is_file()]
    assert index_params.DATA in files
    for p in files:
        assert str(p.relative_to(ROOT)) in tracked, f"{p} is opened by the package but not tracked in git"


def test_nothing_in_the_package_is_ignored_except_build_outputs(tracked):
    r = _git("status", "--ignored", "--porcelain", "--", "uncalled_amd", "include", "tools", "oracle")
    ignored = [l[3:] for l in r.stdout.split("\n") if l.startswith("!! ")]
    allowed = (".so", ".o", ".a", ".hsaco", ".co", ".pyc")
    bad = [p for p in ignored if not (p.endswith(allowed) or p.rstrip("/").endswith(("__pycache__", "_ref", "_build", "variants", "build")))]
    assert not bad, f"ignored non-build files: {bad}"


def test_generators_are_committed(tracked):
    for gen in ("tools/gen_threshs_table.py", "tools/gen_model_table.py", "tests/golden/make_goldens.py"):
        assert gen in tracked


def test_division_by_window_length_is_exact(tmp_path):
    """k_events.hip divides by the window lengths 3 and 6 with a multiply and two FMAs (div_w): the sampled run of the checker
    (every 97th float, 2^22 random double significands at the guard's edge exponents) must agree with the division bit for bit;
    the full run (all floats, 2^33 doubles) is tests/dev/check_div_const.c without an argument."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "check_div_const"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(root / "tests" / "dev" / "check_div_const.c"), "-lm"], check=True)
    out = subprocess.run([str(exe), "quick"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "0 differ" in out.stdout.splitlines()[0] and "0 differ" in out.stdout.splitlines()[1]
