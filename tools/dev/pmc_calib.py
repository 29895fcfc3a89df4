"""Dev tool (GPU box, under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`): known-byte traffic in k_map's access
shape.  n_records x 64 B are written by k_calib_write and read by k_calib_read, `reps` launches each; the counter
totals of those kernels divided by reps * n_records * 64 are the calibration factors tools/dev/summarise_pmc.py applies."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402,F401

from uncalled_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27          # 8 GB of records: far past L2 + Infinity Cache
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = capi.load()
rc = L.unc_calib_traffic(0, n, reps)
print("calib rc", rc, "records", n | 1, "reps", reps, "bytes per launch", (n | 1) * 64)
