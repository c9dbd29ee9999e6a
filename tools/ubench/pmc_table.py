#!/usr/bin/env python3
"""PMC calibration tables of the round-3 microbenchmarks: per dispatch, counters per vector instruction.
usage: pmc_table.py <dir with vc_a vc_b vc_c cl_a cl_c pr_a sub-directories of rocprofv3 --pmc passes>"""
import csv, glob, collections, os, re, sys

def load(d):
    rows = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            rows.setdefault(k, {})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return rows

root = sys.argv[1]
g = lambda r, k: r.get(k, float("nan"))

def names_from(log, pat):
    return [m.group(1).strip() for m in (re.match(pat, l) for l in open(log)) if m]

print("== valu_ceiling pmc: one dispatch per kind, 8 waves per SIMD, 8192 waves x 1 024 000 vector instructions")
print("   windows = 4-cycle issue windows used per instruction = SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2; busy = windows x 4 / measured SIMD cycles per instruction")
a, b, c = load(os.path.join(root, "vc_a")), load(os.path.join(root, "vc_b")), load(os.path.join(root, "vc_c"))
names = names_from(os.path.join(root, "vc_a.log"), r"pmc k<\d+> (.*?)\s+waves ")
print(f"{'kind':52s} {'INSTS_VALU':>11s} {'ACTIVE_VALU':>11s} {'VALU2':>6s} {'windows':>8s} {'cyc/instr':>9s} {'busy':>5s} {'WAIT_INST/WAVE_CYC':>18s} {'SALU/VALU':>9s}")
for i, ((k, ra), (_, rb), (_, rc)) in enumerate(zip(a.items(), b.items(), c.items())):
    iv = ra["SQ_INSTS_VALU"]
    cyc = rc["GRBM_GUI_ACTIVE"] / 8 / (iv / 1024)
    act, v2 = g(ra, "SQ_ACTIVE_INST_VALU") / iv, g(rb, "SQ_ACTIVE_INST_VALU2") / iv
    print(f"{names[i][:52]:52s} {iv / 1e6:10.1f}M {act:11.3f} {v2:6.3f} {act - v2:8.3f} {cyc:9.2f} {(act - v2) * 4 / cyc:5.2f} {ra['SQ_WAIT_INST_ANY'] / ra['SQ_WAVE_CYCLES']:18.2f} {g(rb, 'SQ_INSTS_SALU') / iv:9.2f}")

for tag, title, log, pat in (("cl", "valu_classes pmc (8 waves per SIMD)", "valu_classes.txt", r"(.*?)\s+\| w="), ("pr", "valu_pairs pmc: stream (v_add_f32, K) x 128", "valu_pairs.txt", r"(.*?)\s+\| with")):
    da = os.path.join(root, tag + "_a")
    if not os.path.isdir(da):
        continue
    a = load(da)
    names = names_from(os.path.join(root, log), pat)
    print(f"\n== {title}: share of instructions issued as the SECOND of a window (SQ_ACTIVE_INST_VALU2 / SQ_INSTS_VALU; 0.5 = every window holds two)")
    for i, (k, ra) in enumerate(a.items()):
        iv = ra["SQ_INSTS_VALU"]
        nm = names[i] if i < len(names) else k[1]
        print(f"{nm[:52]:52s} ACTIVE/instr {g(ra, 'SQ_ACTIVE_INST_VALU') / iv:5.2f}  VALU2/instr {g(ra, 'SQ_ACTIVE_INST_VALU2') / iv:5.2f}")
