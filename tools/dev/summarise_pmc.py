"""Dev tool: turn the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/dev/final_run.sh into profiles/<name>.json.

    python tools/dev/summarise_pmc.py gpurun_out/final profiles/r01_pmc_k_map.json 12000
Counter units and caveats as in /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are reported in KiB;
FETCH_SIZE is taken as reported (the guide's 1/2 factor is calibrated for wide coalesced streams only, this kernel gathers
16 B per lane out of 128-byte records and 64-byte FM blocks)."""
import csv
import glob
import json
import sys
from pathlib import Path

src, out, reads = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
tot, launches, dur = {}, {}, {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(str(src / ("pmc_" + name) / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_map" in k and r["Counter_Name"] == name:
                tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
                launches[name] = launches.get(name, set()) | {r.get("Dispatch_Id")}
    for f in glob.glob(str(src / ("pmc_" + name) / "**" / "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_map" in r.get("Kernel_Name", ""):
                dur[name] = dur.get(name, 0.0) + (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
fetch_b = tot.get("FETCH_SIZE", 0.0) * 1024.0
write_b = tot.get("WRITE_SIZE", 0.0) * 1024.0
res = {
    "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --reads %d --steps 1 "
               "--warmup 0 --no-cpu-baseline --no-profile-pass (one pass per counter, tools/dev/final_run.sh)" % reads,
    "kernel": "unc::k_map<false>", "reads_per_launch": reads,
    "counters_KiB": tot, "k_map_launches": {k: len(v) for k, v in launches.items()}, "k_map_ms_under_pmc": dur,
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
    "hbm_bytes_per_read": (fetch_b + write_b) / reads,
    "note": "FETCH_SIZE/WRITE_SIZE in KiB, summed over the k_map dispatches of the run (the main launch plus the few-read re-map "
            "launches for reads whose seed-cluster set outgrew its slot); FETCH_SIZE as reported, see MI355X_MICROARCH.md (HBM).",
}
out.write_text(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
