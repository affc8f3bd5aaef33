"""Generic per-kernel counter table from a rocprofv3 PMC pass (rocpd sqlite): mean of every collected counter per launch.
    python scripts/rocpd_counters.py results.db [name-filter]
"""
import sqlite3
import sys


def main(db_path, filt=""):
    db = sqlite3.connect(db_path)
    agg = {}
    for name, cname, val, dur in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if filt and filt not in name:
            continue
        a = agg.setdefault(name, {})
        c = a.setdefault(cname, [0, 0.0, 0.0])
        c[0] += 1; c[1] += val; c[2] += dur
    for name, a in agg.items():
        short = name.replace("(anonymous namespace)::", "")[:70]
        n = max(v[0] for v in a.values())
        dur = max(v[2] for v in a.values()) / n
        print("%s  launches %d  avg %.1f us" % (short, n, dur / 1e3))
        for cname in sorted(a):
            print("    %-32s %16.0f per launch" % (cname, a[cname][1] / a[cname][0]))


if __name__ == "__main__":
    main(*sys.argv[1:3])
