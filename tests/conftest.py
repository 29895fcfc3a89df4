import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch  # noqa: F401  -- before any test module loads libuncalled_hip.so: one HIP runtime per process (uncalled_amd/__init__.py)

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"
EX_PREFIX = GOLD / "example_index" / "example_ref"


def build_lock():
    """Builds of test infrastructure (oracle, emulator library, host module) are serialised across the workers of a parallel run:
    the first worker builds, the others find the targets up to date."""
    import contextlib
    import fcntl

    @contextlib.contextmanager
    def _lock():
        with open(ROOT / "tests" / ".build.lock", "w") as fh:
            fcntl.flock(fh, fcntl.LOCK_EX)
            try:
                yield
            finally:
                fcntl.flock(fh, fcntl.LOCK_UN)
    return _lock()


def locked_make(*args):
    with build_lock():
        subprocess.run(["make", "-s", *args], check=True)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite is mostly the kernel sources under the emulator, one fiber set per test and single-threaded: without a GPU
    (and unless the command line or UNC_TEST_WORKERS says otherwise) the tests are spread over worker processes (pytest-xdist),
    13 minutes -> 3 on 8 cores.  The GPU suite stays in ONE process: its tests time kernels and share the device."""
    import os
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return None
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return None
    want = os.environ.get("UNC_TEST_WORKERS")
    n = int(want) if want is not None else max(0, min(6, (os.cpu_count() or 1) - 2))
    if n < 2 or _has_gpu():
        return None
    config.option.numprocesses = n
    config.option.dist = "load"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "lanesim: runs the HIP kernel sources under the CPU SIMT emulator")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# the longest emulator cases (a minute or two each): started first, so that a parallel run does not end on one of them alone
SLOW_FIRST = ("test_chunked_mid_reference_team_sort", "test_chunked_stage_tap", "test_matches_small_builder_on_repeats",
              "test_chunked_realtime_path", "test_cluster_pool_pressure", "test_narrow_buckets", "test_parameter_variants")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    items.sort(key=lambda it: 0 if it.name.split("[")[0] in SLOW_FIRST else 1)      # (stable: the rest keeps its order)
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """TEST INFRASTRUCTURE: the plain-C restatement (builds on demand)."""
    from oracle import pyoracle
    if not pyoracle.available():
        locked_make("-C", str(ROOT / "oracle"), "oracle")
    return pyoracle


@pytest.fixture(scope="session")
def ref_lib():
    """TEST INFRASTRUCTURE: the reference's own object code; only where /root/reference exists."""
    from oracle import pyref
    if not pyref.available():
        if not Path("/root/reference/src/mapper.cpp").exists():
            pytest.skip("reference sources not present (GPU box): oracle/_ref was not prebuilt")
        locked_make("-C", str(ROOT / "oracle"), "ref")
    return pyref


@pytest.fixture(scope="session")
def example():
    ex = np.load(GOLD / "example_read.npz")
    return dict(signal=ex["signal"], range=float(ex["range"]), offset=float(ex["offset"]),
                digitisation=float(ex["digitisation"]), prefix=EX_PREFIX)


@pytest.fixture(scope="session")
def goldens():
    return np.load(GOLD / "ref_goldens.npz")


@pytest.fixture(scope="session")
def sim_lib():
    """The product's kernel sources compiled against tests/lanesim (CPU SIMT emulator)."""
    import os
    from uncalled_amd import capi
    # the chunked path's teams: 8 wavefronts per channel are 512 fibers per emulated workgroup -- three times the run time of the
    # chunked cases.  The emulator suite runs them on teams of 2 and has dedicated cases for 4 and 8 (test_chunked_team_sizes); the
    # GPU suite runs everything on the default (8).
    # (UNC_SIM_RT_TEAM is read by the emulator build only: the gfx950 library in the same process keeps its default of 8)
    os.environ.setdefault("UNC_SIM_RT_TEAM", "2")
    extra = os.environ.get("UNC_LANESIM_EXTRA")      # dev: the emulator suite over a variant build (extra -D flags)
    if extra:
        locked_make("-C", str(ROOT / "tests" / "lanesim"), "OUT=_build_extra", "EXTRA=" + extra)
        return capi.load(ROOT / "tests" / "lanesim" / "_build_extra" / "libuncalled_sim.so")
    locked_make("-C", str(ROOT / "tests" / "lanesim"))
    return capi.load(ROOT / "tests" / "lanesim" / "_build" / "libuncalled_sim.so")


@pytest.fixture(scope="session")
def sim_host(sim_lib):
    """TEST INFRASTRUCTURE: the host module's sources (MapPool, RealtimePool, ...) linked against the lanesim build of the
    C ABI, so that their logic runs in the GPU-less container.  Lives beside the lanesim library, never in the package."""
    import importlib.util
    import sysconfig
    import __graft_entry__ as g
    out_dir = ROOT / "tests" / "lanesim" / "_build"
    with build_lock():
        mod = g.build_host(lib=out_dir / "libuncalled_sim.so", out_dir=out_dir)
    assert mod.name == "_uncalled_amd" + sysconfig.get_config_var("EXT_SUFFIX")
    spec = importlib.util.spec_from_file_location("_uncalled_amd", mod)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def hip_lib():
    """The real gfx950 library; GPU tests fail loudly if it is missing."""
    from uncalled_amd import capi
    return capi.load()
