#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc SQ pass to per-kernel averages per launch (see tools/profile_sq.sh).
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(out + "/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        c = r.get("Counter_Name") or r.get("Counter Name")
        v = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
        a = acc[name][c]
        a[0] += v
        a[1] += 1
res = {}
for name, cs in acc.items():
    avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    gui = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                     # kernel duration in shader cycles
    if gui < 1e5:
        continue
    mfma = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    wc = max(avg.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    res[name[:90]] = {
        "launches": int(max(v[1] for v in cs.values())),
        "kernel_cycles": round(gui),
        "mfma_busy_frac_of_1024_simd_cycles": round(mfma / (gui * 1024.0), 4),
        "wave_time_split": {"issuing": round(avg.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                            "of_which_valu_incl_mfma": round(avg.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                            "issue_stalled": round(avg.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                            "parked_waitcnt_barrier": round(avg.get("SQ_WAIT_ANY", 0) / wc, 3)},
        "avg_waves_resident_per_simd": round(wc * 4.0 / (gui * 1024.0), 2),
    }
print(json.dumps(dict(sorted(res.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["launches"])[:10]), indent=1))
