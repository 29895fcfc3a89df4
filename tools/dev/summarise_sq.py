"""Dev tool: rocprofv3 --pmc passes of tools/dev/pmc_sq.sh -> one JSON summary (counter totals over the k_map dispatches of each
pass, the pass's own k_map duration from the SAME directory's kernel trace, derived shares).

    python tools/dev/summarise_sq.py gpurun_out/pmc_sq 50000 [out.json]"""
import csv
import glob
import json
import sys
from pathlib import Path

src, reads = Path(sys.argv[1]), int(sys.argv[2])
out = Path(sys.argv[3]) if len(sys.argv) > 3 else src / "summary.json"
workload = sys.argv[4] if len(sys.argv) > 4 else "ecoli"
tot, passes = {}, {}
for d in sorted(p for p in src.iterdir() if p.is_dir()):
    cc = glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True)
    kt = glob.glob(str(d / "**" / "*kernel_trace.csv"), recursive=True)
    disp, names = set(), set()
    for f in cc:
        for r in csv.DictReader(open(f)):
            if "k_map" in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                disp.add(r.get("Dispatch_Id")); names.add(r["Counter_Name"])
    if not cc:
        continue
    if not kt:
        raise SystemExit(f"{d}: no kernel trace beside the counters (deleted before summarising?)")
    dur = []
    for f in kt:
        if abs(Path(f).stat().st_mtime - Path(cc[0]).stat().st_mtime) > 600:
            raise SystemExit(f"{f} was not written by the pass that wrote {cc[0]}")
        for r in csv.DictReader(open(f)):
            if "k_map" in r.get("Kernel_Name", ""):
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    passes[d.name] = {"counters": sorted(names), "k_map_dispatches": len(disp), "k_map_ms_under_pmc": dur}
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402  (kernel_source_hash: the record is keyed by the kernel it was taken on; bench.py refuses it for another)
res = {"workload": workload, "reads_per_launch": reads, "kernel_source_sha256": bench.kernel_source_hash(),
       "kernel": "unc::k_map<false, false> (64-bit rows, 128-bit keys)" if workload == "grch38" else "unc::k_map<false, true> (32-bit rows)",
       "passes": passes, "counters": tot,
       "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md); one k_map dispatch per pass"}
g = tot.get
der = {}
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    # (SQ_ACTIVE_INST_VMEM read 0.0 in every pass of rounds 3-4 -- a dead counter on this build -- and is no longer reported)
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
        if g(k) is not None:
            der["wave_cycle_share_" + k[3:].lower()] = g(k) / wc
    if g("SQ_INST_LEVEL_VMEM") is not None:
        # vector-memory instructions in flight, summed over cycles: per wave-cycle = how many a wavefront has outstanding on average
        # while it is resident -- the direct witness of "waiting on memory" (SQ_WAIT_ANY counts every kind of wait)
        der["vmem_in_flight_per_wave_cycle"] = g("SQ_INST_LEVEL_VMEM") / (4.0 * wc)
        der["vmem_in_flight_note"] = "SQ_INST_LEVEL_VMEM / (4 x SQ_WAVE_CYCLES): the level counts per cycle, the wave cycles per quad-cycle"
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT",
          "SQ_INSTS_BRANCH", "SQ_INSTS", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"):
    if g(k) is not None:
        der[k[3:].lower() + "_per_read"] = g(k) / reads
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INST_CYCLES_VALU"):
    der["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INST_CYCLES_VALU"))
# how busy the vector ALUs are (round 5 took a wave64 VALU instruction to occupy its SIMD for 4 cycles and read 4 x SQ_ACTIVE_INST_VALU
# as VALU-busy: 55 - 60 %; round 6 measured two cycles -- see below)
for name, p in passes.items():
    if "SQ_ACTIVE_INST_VALU" in p["counters"] and p["k_map_ms_under_pmc"]:
        ms = sum(p["k_map_ms_under_pmc"])
        # round 6: a wave64 VALU instruction holds its SIMD for TWO cycles (tools/dev/ubench_valu.hip); SQ_ACTIVE_INST_VALU counts
        # quad-cycles and rounds every instruction up to one, so 4 x that counter is an upper bound (kept as valu_quad_cycle_share)
        der["valu_quad_cycle_share"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024 * 2.4e9 * ms * 1e-3)
        if g("SQ_INSTS_VALU") is not None:
            der["valu_pipe_busy"] = 2.0 * g("SQ_INSTS_VALU") / (1024 * 2.4e9 * ms * 1e-3)
# issue utilisation: wave-instructions / (1024 SIMDs x clock x k_map's duration in the pass that counted them); the clock is
# the 2.4 GHz peak engine clock (MI355X_MICROARCH.md), so this is a lower bound on the share of issue slots used
CLOCK_HZ, SIMDS = 2.4e9, 1024
for name, p in passes.items():
    if "SQ_INSTS" in p["counters"] and p["k_map_ms_under_pmc"]:
        ms = sum(p["k_map_ms_under_pmc"])
        der["issue_utilisation"] = g("SQ_INSTS") / (SIMDS * CLOCK_HZ * ms * 1e-3)
        der["issue_utilisation_note"] = f"SQ_INSTS / (1024 SIMDs x 2.4 GHz x {ms:.1f} ms of k_map under the counters)"
    if "SQ_INSTS_VALU" in p["counters"] and p["k_map_ms_under_pmc"]:
        ms = sum(p["k_map_ms_under_pmc"])
        der["valu_issue_utilisation"] = g("SQ_INSTS_VALU") / (SIMDS * CLOCK_HZ * ms * 1e-3)
res["derived"] = der
out.write_text(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
