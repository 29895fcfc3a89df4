#!/usr/bin/env python3
"""Regenerates tests/golden/example_read.npz and tests/golden/example_index/ from the
reference's bundled example (/root/reference/example).  Runs only in the
build container (needs /root/reference and /opt/conda/bin/h5dump); the outputs are committed
because /root/reference does not exist on the GPU box.

The int16 samples are read with h5dump (no h5py here).  Calibration: the fast5 stores
range=1534.141357421875 (a float32 widened to double), offset=10, digitisation=8192; the
reference routes the attributes through strings (read_buffer.cpp:213-222) and the survey's
probe used the 6-significant-digit rendering 1534.14, so that is what the fixture pins.
"""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/example")
H5DUMP = "/opt/conda/bin/h5dump"
OUT = Path(__file__).resolve().parent


def h5attr(f, path):
    txt = subprocess.run([H5DUMP, "-a", path, str(f)], check=True, capture_output=True, text=True).stdout
    return re.search(r"\(0\): (.*)", txt).group(1).strip().strip('"')


def main():
    f5 = next(REF.glob("*.fast5"))
    txt = subprocess.run([H5DUMP, "-d", "/Raw/Reads/Read_101/Signal", "-y", "-w", "0", str(f5)],
                         check=True, capture_output=True, text=True).stdout
    body = txt[txt.index("DATA {") + 6: txt.rindex("}\n}\n}")]
    sig = np.array([int(x) for x in body.replace("\n", " ").split(",") if x.strip()], dtype=np.int16)
    assert sig.size == 31668, sig.size
    np.savez_compressed(
        OUT / "example_read.npz",
        signal=sig,
        read_id=h5attr(f5, "/Raw/Reads/Read_101/read_id"),
        read_number=np.int64(h5attr(f5, "/Raw/Reads/Read_101/read_number")),
        start_time=np.int64(h5attr(f5, "/Raw/Reads/Read_101/start_time")),
        channel=np.int64(h5attr(f5, "/UniqueGlobalKey/channel_id/channel_number")),
        range=np.float64(1534.14),
        offset=np.float64(h5attr(f5, "/UniqueGlobalKey/channel_id/offset")),
        digitisation=np.float64(h5attr(f5, "/UniqueGlobalKey/channel_id/digitisation")),
    )
    idx = OUT / "example_index"
    idx.mkdir(exist_ok=True)
    for p in (REF / "index").iterdir():
        shutil.copyfile(p, idx / p.name)
    shutil.copyfile(REF / "example_ref.fa", idx / "example_ref.fa")
    print("wrote", OUT / "example_read.npz", "and", idx)


if __name__ == "__main__":
    sys.exit(main())
