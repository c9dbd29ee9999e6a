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
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]] += 1
for k in sums:
    print("== PMC per dispatch (mean):", k)
    for c in sorted(sums[k]):
        print(f"   {c:28s} {sums[k][c] / calls[k][c]:18.1f}   (n={calls[k][c]})")
