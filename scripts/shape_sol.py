"""Per-shape speed-of-light table from a `bench.py --per-shape` log: for every conv launch shape, measured ms/step
against max(FLOPs / 2.5 PFLOP/s, ideal bytes / 6.3 TB/s achievable)."""
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\[conv\] (\w+) \((\d+), (\d+), (\d+), (\d+)\) x \((\d+), (\d+)(?:, (\d+), (\d+))?\) k(\d) s(\d)\s+n=\s*(\d+)\s+([\d.]+) ms/step\s+([\d.]+) TFLOP", l)
    if not m:
        continue
    kind = m.group(1)
    n, h, w, c = map(int, m.group(2, 3, 4, 5))
    k, s, cnt = int(m.group(10)), int(m.group(11)), int(m.group(12))
    ms, tf = float(m.group(13)), float(m.group(14))
    flops = tf * 1e12 * ms * 1e-3 / cnt
    if kind == "wgrad":
        n2, h2, w2, c2 = int(m.group(6)), int(m.group(7)), int(m.group(8)), int(m.group(9))
        byt = 2 * (n * h * w * c + n2 * h2 * w2 * c2) + 4 * c * c2 * k * k
    elif kind == "fwd":
        cout = int(m.group(6)); ho = h // s
        byt = 2 * (n * h * w * c + n * ho * ho * cout)
    else:
        cin = int(m.group(6))
        byt = 2 * (n * h * w * c + n * h * s * w * s * cin)
    t_m, t_h = flops / 2.5e15 * 1e3, byt / 6.3e12 * 1e3
    rows.append((kind, (n, h, w, c), k, s, cnt, ms, cnt * max(t_m, t_h), "mfma" if t_m > t_h else "hbm", l.split("x (")[1].split(")")[0]))
agg = {}
for r in rows:
    a = agg.setdefault((r[0], r[7]), [0.0, 0.0])
    a[0] += r[5]; a[1] += r[6]
for k, v in sorted(agg.items()):
    print("%-6s %-5s measured %6.3f ms  sol %6.3f ms  ratio %.2f" % (k[0], k[1], v[0], v[1], v[0] / max(v[1], 1e-9)))
if len(sys.argv) > 2:
    for r in sorted(rows, key=lambda r: -(r[5] - r[6]))[:int(sys.argv[2])]:
        print("%-6s %-22s x(%-14s) k%d s%d n=%d  %.3f ms  sol %.3f  %s" % (r[0], r[1], r[8], r[2], r[3], r[4], r[5], r[6], r[7]))
