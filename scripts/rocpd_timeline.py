"""Per-queue timeline of ONE training step from a rocprofv3 kernel trace (rocpd SQLite database):
busy time, idle time and launch count of every HIP stream (hardware queue), and the largest idle gaps of the
busiest queue with the kernels on either side.
    python scripts/rocpd_timeline.py gpurun_out/prof/x_results.db [step_index_from_end]
A step is delimited by the sgd_kernel launches."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 0: the step of median length (a profiler flush can stall one step by milliseconds)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
if back <= 0:
    spans = []
    for i in range(1, len(sgd)):
        seg = rows[sgd[i - 1] + 1:sgd[i] + 1]
        spans.append((max(r[2] for r in seg) - seg[0][1], i))
    spans = sorted(spans[len(spans) // 3:])               # (the first third are warm-up steps)
    mid = spans[len(spans) // 2][1]
    lo, hi = sgd[mid - 1] + 1, sgd[mid] + 1
else:
    lo, hi = sgd[-back - 1] + 1, sgd[-back] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
print("step: %d launches, %.3f ms wall (%s column: %s)" % (len(step), (t1 - t0) / 1e6, "queue", qcol))
by = {}
for r in step:
    by.setdefault(r[3], []).append(r)
main_q = max(by, key=lambda q: sum(r[2] - r[1] for r in by[q]))
for q, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in rs)
    span = max(r[2] for r in rs) - min(r[1] for r in rs)
    gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print("queue %s: %4d launches  busy %7.3f ms  first..last %7.3f ms  idle inside %7.3f ms  (gaps < 20 us: %d, sum %.3f ms; >= 20 us: %d, sum %.3f ms)" % (
        q, len(rs), busy / 1e6, span / 1e6, sum(pos) / 1e6, sum(1 for g in pos if g < 20000), sum(g for g in pos if g < 20000) / 1e6,
        sum(1 for g in pos if g >= 20000), sum(g for g in pos if g >= 20000) / 1e6))
rs = by[main_q]
gaps = sorted(((rs[i + 1][1] - rs[i][2], i) for i in range(len(rs) - 1)), reverse=True)[:15]
print("largest idle gaps on queue %s:" % main_q)
for g, i in gaps:
    print("  %7.1f us at +%.3f ms  after %-60s before %s" % (g / 1e3, (rs[i][2] - t0) / 1e6, rs[i][0][:60].replace("(anonymous namespace)::", ""), rs[i + 1][0][:60].replace("(anonymous namespace)::", "")))
# union busy time over all queues (any kernel running)
ev = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
cur, last, anybusy = 0, t0, 0
for t, d in ev:
    if cur > 0:
        anybusy += t - last
    cur += d; last = t
print("GPU busy with at least one kernel: %.3f ms of %.3f ms" % (anybusy / 1e6, (t1 - t0) / 1e6))
if len(sys.argv) > 3:
    # full listing of the step, every queue, in start order:  +ms  queue  duration  name
    with open(sys.argv[3], "w") as f:
        for r in step:
            f.write("%8.3f q%-2s %7.1f us  %s\n" % ((r[1] - t0) / 1e6, r[3], (r[2] - r[1]) / 1e3, r[0][:110].replace("(anonymous namespace)::", "")))
