"""The scheduler's and the node pool's rings (uncalled_amd/csrc/map_sched.h) compiled for gfx950: no cache-wide operation inside them.

On this chip an agent-scope ACQUIRE is `buffer_inv sc1` (an invalidate of the XCD's whole L2) and a RELEASE `buffer_wbl2 sc1` (a write-back
of it).  Rounds 2-5 polled the rings with acquire loads; round 6 found whole XCDs held up in such a loop for the length of a GRCh38
launch.  The rings now carry value and sequence number in one 64-bit word and use relaxed atomics only; the one
release / acquire per hand-over is the caller's, outside any loop.  This test reads the ISA (hipcc cross-compiles without a GPU)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "uncalled_amd" / "csrc"

SRC = r'''
#include <hip/hip_runtime.h>
#include "unc_dev_types.h"
#include "wave_prims.h"
#include "map_sched.h"
using namespace unc;
extern "C" __global__ void t_sched_pop(SchedQueue *q, SchedCell *c, uint32_t mask, uint32_t *out) { out[0] = sched_pop(q, c, mask); }
extern "C" __global__ void t_sched_push(SchedQueue *q, SchedCell *c, uint32_t mask, uint32_t v) { sched_push(q, c, mask, v); }
extern "C" __global__ void t_pool_pop(PoolQueue *q, SchedCell *c, uint32_t mask, uint32_t *out) { out[0] = pool_ring_pop(q, c, mask); }
extern "C" __global__ void t_pool_push(PoolQueue *q, SchedCell *c, uint32_t mask, uint32_t v) { pool_ring_push(q, c, mask, v); }
'''


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    pytest.skip("hipcc not found")


def test_rings_hold_no_cache_wide_operation(tmp_path):
    src = tmp_path / "rings.hip"
    src.write_text(SRC)
    asm = tmp_path / "rings.s"
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", str(CSRC), str(src), "-o", str(asm)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    bodies = {}
    for name in ("t_sched_pop", "t_sched_push", "t_pool_pop", "t_pool_push"):
        m = re.search(r"^%s:.*?s_endpgm" % name, text, re.S | re.M)
        assert m, name
        bodies[name] = m.group(0)
    for name, body in bodies.items():
        assert "buffer_inv" not in body and "buffer_wbl2" not in body, (name, [l for l in body.splitlines() if "buffer_" in l])
        # a cell is ONE 64-bit access that goes to the memory side (sc1): value and sequence number cannot be seen apart
        assert re.search(r"global_load_dwordx2 .* sc1", body), name
        assert re.search(r"global_store_dwordx2 .* sc1", body), name
    # the pool's pop has no compare-and-swap (nothing another wavefront's progress can restart); the scheduler's non-blocking pop has one
    assert "cmpswap" not in bodies["t_pool_pop"] and "cmpswap" not in bodies["t_pool_push"] and "cmpswap" not in bodies["t_sched_push"]
    assert "cmpswap" in bodies["t_sched_pop"]
