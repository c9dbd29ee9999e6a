#!/usr/bin/env python3
"""Summarises a tools/prof.sh output directory: kernel-trace stats + PMC sums per kernel."""
import csv, glob, os, sys, collections

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:3]:
        print("dispatch:", {k: r[k] for k in r if k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size", "Start_Timestamp", "End_Timestamp")})
sums = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
each = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:80]
        sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
        each[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        calls[k][row["Counter_Name"]] += 1
for k in sums:
    print("== PMC per dispatch (mean):", k)
    for c in sorted(sums[k]):
        print(f"   {c:28s} {sums[k][c] / calls[k][c]:18.1f}   (n={calls[k][c]})")

# HBM traffic of the trace kernel per launch from the TCC memory-side request counters.
# Calibration (MI355X_MICROARCH.md, HBM): a read request is a 128-B line tallied at 64 B, i.e. bytes read
# = RDREQ x 128 (here: 8 float4 pixels per line -> W*H/8 requests, which matches the count exactly);
# a write request is 64 B.
# The launch WITHOUT the cost measurement is reported (the lower median: the bench command's profile pass holds two launches
# of the kernel, the first of which - like one launch in 32 - also measures the per-tile cost, one atomic per pixel that
# reaches the memory side as a write request); the measuring launch's figure is listed as "cost_measuring_launch".
import json, re, statistics
for k in sums:
    if re.search(r"rt_path_kernel<(true|false), false", k) and "TCC_EA0_RDREQ_sum" in sums[k]:
        rd = statistics.median_low(each[k]["TCC_EA0_RDREQ_sum"])
        wr = statistics.median_low(each[k]["TCC_EA0_WRREQ_sum"])
        t = {"kernel": k, "rdreq_per_launch": rd, "wrreq_per_launch": wr, "bytes_per_launch": rd * 128 + wr * 64,
             "cost_measuring_launch_bytes": max(each[k]["TCC_EA0_RDREQ_sum"]) * 128 + max(each[k]["TCC_EA0_WRREQ_sum"]) * 64,
             "launches": len(each[k]["TCC_EA0_RDREQ_sum"]), "frames_per_launch": int(os.environ.get("PROF_FRAMES_PER_LAUNCH", "1")),
             "bytes_per_pass": (rd * 128 + wr * 64) / int(os.environ.get("PROF_FRAMES_PER_LAUNCH", "1")),
             "rule": "RDREQ x 128 B + WRREQ x 64 B (separate --pmc pass, TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum), the launch without the cost measurement (lower median)"}
        json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
        print("traffic:", t)

# VALUBusy of the path kernel (profiles/r3_valu_ceiling.txt): windows = SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2 (one 4-cycle
# issue window per instruction, minus the instructions that rode as the second of a window), against SIMDs x shader cycles
# (GRBM_GUI_ACTIVE is summed over the 8 XCDs). bench.py quotes this file as `roofline.valu_busy` with its source.
for k in sums:
    need = ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU")
    if re.search(r"rt_path_kernel<(true|false), false", k) and all(n in sums[k] for n in need):
        m = {n: statistics.median(each[k][n]) for n in need}
        simds = int(os.environ.get("PROF_SIMDS", "1024"))  # 256 CUs x 4
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        v = {"kernel": k, "valu_busy": round(4.0 * (m["SQ_ACTIVE_INST_VALU"] - m["SQ_ACTIVE_INST_VALU2"]) / (simds * cycles), 4),
             "valu2_share": round(m["SQ_ACTIVE_INST_VALU2"] / m["SQ_INSTS_VALU"], 4),
             "lane_utilisation": round(m["SQ_THREAD_CYCLES_VALU"] / m["SQ_INSTS_VALU"] / 64.0, 4),
             "cycles_per_valu_instruction_and_simd": round(simds * cycles / m["SQ_INSTS_VALU"], 3),
             "shader_cycles_per_launch": cycles, "frames_per_launch": int(os.environ.get("PROF_FRAMES_PER_LAUNCH", "1")),
             "rule": "4 x (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), separate --pmc passes, median launch"}
        json.dump(v, open(os.path.join(out, "valu_busy.json"), "w"), indent=1)
        print("valu_busy:", v)
