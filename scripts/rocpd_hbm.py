"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one counter per pass as
MI355X_MICROARCH.md prescribes), with the gfx950 correction: FETCH_SIZE reports half the bytes of wide
coalesced reads, so fetch is doubled; WRITE_SIZE is exact.  Output: CSV per kernel, per bench step.
    python scripts/rocpd_hbm.py fetch.db write.db STEPS_IN_RUN out.csv
"""
import csv
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, val, dur in db.execute("select kernel_name, value, duration from counters_collection where counter_name = ?",
                                     (counter,)):
        a = out.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += val; a[2] += dur
    return out


def main(fdb, wdb, steps, out_path):
    steps = float(steps)
    f = per_kernel(fdb, "FETCH_SIZE")
    w = per_kernel(wdb, "WRITE_SIZE")
    rows = []
    for name in sorted(set(f) | set(w)):
        nf, kbf, durf = f.get(name, [0, 0.0, 0.0])
        nw, kbw, _ = w.get(name, [0, 0.0, 0.0])
        n = max(nf, nw)
        rd = 2.0 * kbf * 1024 / steps
        wr = kbw * 1024 / steps
        ms = durf / steps / 1e6
        rows.append([name, round(n / steps, 1), round(kbf / steps, 1), round(kbw / steps, 1), round((rd + wr) / 1e6, 1),
                     round((rd + wr) / max(n / steps, 1e-9) / 1e6, 2), round(ms, 3),
                     round((rd + wr) / 1e6 / max(ms, 1e-9), 0)])
    rows.sort(key=lambda r: -r[4])
    with open(out_path, "w", newline="") as fh:
        wr_ = csv.writer(fh)
        wr_.writerow(["kernel", "launches_per_step", "FETCH_SIZE_KB_per_step", "WRITE_SIZE_KB_per_step",
                      "hbm_MB_per_step(fetch x2 + write)", "hbm_MB_per_launch", "kernel_ms_per_step(serialised by the counter pass)",
                      "GB_per_s"])
        wr_.writerows(rows)
        wr_.writerow(["TOTAL", "", "", "", round(sum(r[4] for r in rows), 1), "", round(sum(r[6] for r in rows), 3), ""])


if __name__ == "__main__":
    main(*sys.argv[1:5])
