#!/usr/bin/env python3
"""Per-kernel durations AND inter-kernel gaps of the decode step from a rocprofv3 rocpd .db
(kernel-trace).  Groups this library's kernels by (name, grid) so wo and w2 (same kernel) are
separated, and reports the idle gap that follows each kernel inside the hipGraph replay."""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)
ours = [(short(n), s, e, g, l) for (n, s, e, g, l) in rows if re.match(r"^(void )?k_[a-z0-9_]+", n)]
dur = defaultdict(list)
gap = defaultdict(list)
for i, (n, s, e, g, l) in enumerate(ours):
    key = f"{n} grid={g // max(l, 1)}x{l}"
    dur[key].append((e - s) / 1e3)
    if i + 1 < len(ours):
        gp = (ours[i + 1][1] - e) / 1e3
        if gp < 50:  # same replay train
            gap[key].append(gp)
print(f"{'kernel':48s} {'n':>6s} {'avg_us':>8s} {'p50':>8s} {'min':>8s} {'gap_after_avg':>14s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sorted(dur[k])
    g = gap.get(k, [0])
    print(f"{k:48s} {len(d):6d} {sum(d)/len(d):8.2f} {d[len(d)//2]:8.2f} {d[0]:8.2f} {sum(g)/len(g):14.2f}")
