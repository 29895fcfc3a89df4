"""Dev tool: the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/dev/final_run.sh (k_map on the bench's own batch + the
known-byte calibration kernels in the same access shape) -> profiles/<name>.json.

    python tools/dev/summarise_pmc.py <dir of the passes> profiles/rNN_pmc_k_map.json 50000 ecoli [fetch write calib_fetch calib_write]
(the last four: the pass directories under <dir>; tools/dev/pmc_sq.sh names them f w cf cw).  Durations come from the kernel
trace of the SAME pass directory; a trace older than its pass's counters, or a pass without one, is an error.
Both counters are reported in KiB.  The calibration kernels move exactly n_records x 64 B per launch (one lane per 64-byte
record, four 16-byte accesses per lane, scattered over 8 GB): counter / known bytes is the factor k_map's counters are divided
by (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is known to under-report wide streams by 2x, other shapes are uncalibrated)."""
import csv
import glob
import json
import re
import sys
from pathlib import Path

src, out, reads, workload = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
D_FETCH, D_WRITE, D_CF, D_CW = (sys.argv[5:9] if len(sys.argv) >= 9 else ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "calib_FETCH_SIZE", "calib_WRITE_SIZE"))


def collect(dirname, counter, kernel_sub):
    tot, disp = 0.0, set()
    for f in glob.glob(str(src / dirname / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r.get("Kernel_Name", "") and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                disp.add(r.get("Dispatch_Id"))
    return tot * 1024.0, len(disp)


def durations(dirname, kernel_sub):
    """k_map's durations in the kernel trace written by the same rocprofv3 pass as the counters (same directory, not older)"""
    cc = glob.glob(str(src / dirname / "**" / "*counter_collection.csv"), recursive=True)
    kt = glob.glob(str(src / dirname / "**" / "*kernel_trace.csv"), recursive=True)
    if not kt:
        raise SystemExit(f"{src / dirname}: no kernel trace beside the counters (deleted before summarising?)")
    d = []
    for f in kt:
        if cc and abs(Path(f).stat().st_mtime - Path(cc[0]).stat().st_mtime) > 600:
            raise SystemExit(f"{f} was not written by the pass that wrote {cc[0]}")
        for r in csv.DictReader(open(f)):
            if kernel_sub in r.get("Kernel_Name", ""):
                d.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    return d


sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402  (kernel_source_hash: the record is keyed by the kernel it was taken on; bench.py refuses it for another)
res = {"workload": workload, "reads_per_launch": reads, "kernel_source_sha256": bench.kernel_source_hash(),
       "kernel": "unc::k_map<false, false> (64-bit rows, 128-bit keys)" if workload == "grch38" else "unc::k_map<false, true> (32-bit rows)",
       "command": f"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace --output-format csv -- python tools/dev/ab_libs.py {reads}:{workload} "
                  "uncalled_amd/libuncalled_hip.so (the bench's batch: same index, same reads, one k_map dispatch; one pass per counter; "
                  "tools/dev/pmc_sq.sh f w cf cw)"}
calib = {}
for c, kern in (("FETCH_SIZE", "k_calib_read"), ("WRITE_SIZE", "k_calib_write")):
    log = src / ((D_CF if c == "FETCH_SIZE" else D_CW) + ".log")
    known = None
    if log.exists():
        m = re.search(r"bytes per launch (\d+)", log.read_text())
        known = int(m.group(1)) if m else None
    b, n = collect(D_CF if c == "FETCH_SIZE" else D_CW, c, kern)
    calib[c] = {"kernel": kern, "launches": n, "counter_bytes_per_launch": b / n if n else None, "known_bytes_per_launch": known,
                "factor": (b / n / known) if (n and known) else None}
res["calibration"] = calib
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dn = D_FETCH if c == "FETCH_SIZE" else D_WRITE
    b, n = collect(dn, c, "k_map")
    raw[c] = {"counter_bytes": b, "k_map_dispatches": n, "k_map_ms_under_pmc": durations(dn, "k_map")}
res["raw"] = raw
fac_f = calib["FETCH_SIZE"]["factor"] or 1.0
fac_w = calib["WRITE_SIZE"]["factor"] or 1.0
res["fetch_bytes_per_launch"] = raw["FETCH_SIZE"]["counter_bytes"] / fac_f
res["write_bytes_per_launch"] = raw["WRITE_SIZE"]["counter_bytes"] / fac_w
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
res["hbm_bytes_per_read"] = res["hbm_bytes_per_launch"] / reads
res["note"] = ("counter totals over the k_map dispatches of one 50 k-read step, each divided by the factor its calibration kernel "
               "measured in the same access shape (factor 1.0 when the calibration pass is missing: then the figure is the raw counter)")
out.write_text(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
