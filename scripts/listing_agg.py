"""Aggregate a step listing (scripts/rocpd_timeline.py) by kernel and queue:  python scripts/listing_agg.py listing.txt [min_us]"""
import collections
import re
import sys
agg = collections.defaultdict(lambda: [0, 0.0])
tot = {}
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) q(\d)\s+([\d.]+) us\s+(.*)', l)
    if not m:
        continue
    q, us, name = int(m.group(2)), float(m.group(3)), m.group(4)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:70]
    agg[(q, name)][0] += 1
    agg[(q, name)][1] += us
    tot[q] = tot.get(q, 0) + us
for q in sorted(tot):
    print("queue", q, "busy %.2f ms" % (tot[q] / 1e3))
    for (qq, n), v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if qq == q and v[1] > thr:
            print("   %-72s %3d  %8.1f us" % (n, v[0], v[1]))
