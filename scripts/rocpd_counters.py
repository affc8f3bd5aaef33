"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (rocpd database), per bench step, with the ratio of the first two:
    python scripts/rocpd_counters.py db STEPS out.csv
e.g. --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE: the share of LDS-array cycles that are bank-conflict replays."""
import csv
import sqlite3
import sys


def main(db_path, steps, out_path):
    steps = float(steps)
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select distinct counter_name from counters_collection order by counter_name")]
    agg = {}
    for kname, cname, val, dur in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        a = agg.setdefault(kname, {"n": {}, "dur": 0.0})
        a[cname] = a.get(cname, 0.0) + val
        a["n"][cname] = a["n"].get(cname, 0) + 1
        if cname == names[0]:
            a["dur"] += dur
    rows = []
    for k, a in agg.items():
        v = [a.get(c, 0.0) / steps for c in names]
        ratio = v[0] / v[1] if len(v) > 1 and v[1] else 0.0
        rows.append([k.replace("(anonymous namespace)::", "")[:120], round(max(a["n"].values()) / steps, 1), round(a["dur"] / steps / 1e6, 3)] +
                    [round(x, 1) for x in v] + [round(ratio, 4)])
    rows.sort(key=lambda r: -r[2])
    with open(out_path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches_per_step", "kernel_ms_per_step"] + names + ["%s / %s" % (names[0], names[1]) if len(names) > 1 else "ratio"])
        w.writerows(rows)


if __name__ == "__main__":
    main(*sys.argv[1:4])
