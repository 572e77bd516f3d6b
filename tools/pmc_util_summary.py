#!/usr/bin/env python3
"""Per-kernel clock (upper bound: GUI_ACTIVE also covers ~10 us of dispatch overhead per launch) and matrix-pipe utilisation from a `rocprofv3 -i tools/pmc_util.txt --kernel-trace
--output-format csv` directory: clock = GRBM_GUI_ACTIVE / kernel duration, MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE x CUs x 4 SIMDs)."""
import csv, collections, glob, sys
d = sys.argv[1]
CUS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
XCDS = 8
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"{'kernel':64s} {'n':>4s} {'us(pmc run)':>11s} {'clock GHz':>9s} {'MfmaUtil %':>10s} {'VALU busy %':>11s} {'wave wait %':>11s} {'bf16 MFMA TF':>12s}")
for k, v in sorted(cnt.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" not in a or not dur.get(k):
        continue
    us = sum(dur[k]) / len(dur[k]) / 1e3
    if us < 20:
        continue
    gui = a["GRBM_GUI_ACTIVE"] / XCDS                       # the counter is summed over the XCDs
    mfma = 100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * CUS * 4)
    valu = 100 * a.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (gui * CUS * 4) if "SQ_ACTIVE_INST_VALU" in a else float("nan")
    wait = 100 * a.get("SQ_WAIT_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)
    tf = a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) * 512 / (us * 1e-6) / 1e12
    print(f"{k[:64]:64s} {len(dur[k]):4d} {us:11.1f} {gui / (us * 1e3):9.3f} {mfma:10.1f} {valu:11.1f} {wait:11.1f} {tf:12.0f}")
