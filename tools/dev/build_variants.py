"""Dev tool: build variant libraries (extra -D flags) into uncalled_amd/variants/ for tools/dev/ab_libs.py.

    python tools/dev/build_variants.py name1="-DUNC_LB=4 -DX" name2="..." """
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g

out = ROOT / "uncalled_amd" / "variants"
out.mkdir(exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    lib = out / ("libunc_%s.so" % name)
    cmd = [g._hipcc()] + g.HIPCC_FLAGS + flags.split() + [str(g.CSRC / s) for s in g.HIP_SOURCES] + ["-o", str(lib)]
    procs.append((name, subprocess.Popen(cmd, cwd=str(g.CSRC), stderr=subprocess.PIPE, text=True)))
for name, p in procs:
    err = p.communicate()[1]
    print(name, "ok" if p.returncode == 0 else "FAILED\n" + err[-2000:])
