"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs)   (gfx94x formula, guide section PMC)
    python scripts/rocpd_mfma.py results.db STEPS_IN_RUN out.csv
"""
import csv
import sqlite3
import sys


def main(db_path, steps, out_path):
    steps = float(steps)
    db = sqlite3.connect(db_path)
    agg = {}
    for name, cname, val, dur in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        a = agg.setdefault(name, {"n": 0, "mfma": 0.0, "gui": 0.0, "dur": 0.0})
        if cname == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["mfma"] += val; a["n"] += 1; a["dur"] += dur
        elif cname == "GRBM_GUI_ACTIVE":
            a["gui"] += val
    rows = []
    for name, a in agg.items():
        if a["mfma"] <= 0:
            continue
        denom = a["gui"] / 8.0 * 256 * 4
        clock = a["gui"] / 8.0 / max(a["dur"], 1e-9)          # GHz (cycles per ns), reads high on short dispatches
        rows.append([name, round(a["n"] / steps, 1), round(a["dur"] / steps / 1e6, 3), round(100.0 * a["mfma"] / max(denom, 1.0), 1),
                     round(clock, 2)])
    rows.sort(key=lambda r: -r[2])
    with open(out_path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches_per_step", "kernel_ms_per_step(serialised)", "MfmaUtil_percent", "GRBM_GUI_ACTIVE/8/duration_GHz"])
        w.writerows(rows)


if __name__ == "__main__":
    main(*sys.argv[1:4])
