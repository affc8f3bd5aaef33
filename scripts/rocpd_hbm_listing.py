"""Per-DISPATCH HBM traffic of one bench step from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; same corrections as
rocpd_hbm.py): the launches of the last step in launch order, so that a launch whose traffic is well above its algorithmic
bytes can be named (per-kernel averages hide it when one kernel serves several shapes).
    python scripts/rocpd_hbm_listing.py fetch.db write.db out.txt
A step ends with sgd_kernel; the last complete step of each pass is taken and the two passes are joined by position."""
import sqlite3
import sys


def dispatches(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
    grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    wg = "workgroup_size" if "workgroup_size" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
    q = "select kernel_name, value, duration, %s, %s from counters_collection where counter_name = ? order by %s" % (grid, wg, order)
    rows = list(db.execute(q, (counter,)))
    ends = [i for i, r in enumerate(rows) if r[0].startswith("(anonymous namespace)::sgd_kernel") or "sgd_kernel(" in r[0]]
    if len(ends) < 2:
        raise SystemExit("fewer than two steps in " + db_path)
    return rows[ends[-2] + 1: ends[-1] + 1]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "u16").replace("unsigned int", "u32")


def main(fdb, wdb, out_path):
    # "-" for a pass that was not collected: its column reads 0
    w = dispatches(wdb, "WRITE_SIZE") if wdb != "-" else None
    f = dispatches(fdb, "FETCH_SIZE") if fdb != "-" else [(r[0], 0.0, r[2], r[3], r[4]) for r in w]
    if w is None:
        w = [(r[0], 0.0, r[2], r[3], r[4]) for r in f]
    if len(f) != len(w) or any(a[0] != b[0] for a, b in zip(f, w)):
        raise SystemExit("the two passes do not hold the same launch sequence (%d vs %d)" % (len(f), len(w)))
    tot_r = tot_w = 0.0
    with open(out_path, "w") as fh:
        fh.write("#   i  read MB (FETCH_SIZE x2)  write MB   us (serialised)  blocks  kernel\n")
        for i, (a, b) in enumerate(zip(f, w)):
            rd, wr = 2.0 * a[1] * 1024 / 1e6, b[1] * 1024 / 1e6
            tot_r += rd; tot_w += wr
            blocks = int(a[3]) // max(int(a[4]), 1) if a[3] else 0
            fh.write("%5d %10.1f %10.1f %10.1f %7d  %s\n" % (i, rd, wr, a[2] / 1e3, blocks, short(a[0])[:110]))
        fh.write("# total read %.1f MB  write %.1f MB  sum %.1f MB over %d launches\n" % (tot_r, tot_w, tot_r + tot_w, len(f)))


if __name__ == "__main__":
    main(*sys.argv[1:4])
