"""Kernel-stats summary (the table `rocprofv3 --stats` prints) from a rocprofv3 rocpd SQLite database:
    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db profiles/name.csv
"""
import csv
import sqlite3
import statistics
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, end - start from kernels").fetchall()
    by = {}
    for name, d in rows:
        by.setdefault(name, []).append(d)
    total = float(sum(sum(v) for v in by.values()))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 3)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
