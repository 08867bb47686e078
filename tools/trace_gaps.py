#!/usr/bin/env python3
"""Idle time of the GPU inside a rocprofv3 --kernel-trace run: union of the kernels' [start, end] intervals against the span they cover,
and the largest gaps with the kernels on either side.  python tools/trace_gaps.py <results.db> [skip_first_n_kernels] [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(db.execute(f"select {name_col}, start, end from {view} order by start"))[skip:]


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]


busy, cur_s, cur_e, gaps = 0, rows[0][1], rows[0][2], []
prev_name = rows[0][0]
for name, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, short(prev_name), short(name)))
        cur_s, cur_e = s, e
        prev_name = name
    elif e > cur_e:
        cur_e = e
        prev_name = name
busy += cur_e - cur_s
span = rows[-1][2] - rows[0][1]
print(f"kernels {len(rows)}; span {span / 1e6:.2f} ms; busy (union) {busy / 1e6:.2f} ms = {busy / span:.3f}; idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
agg = {}
for g, a, b in gaps:
    k = (a, b)
    t = agg.setdefault(k, [0, 0])
    t[0] += g
    t[1] += 1
for (a, b), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {t / 1e6:8.3f} ms in {c:5d} gaps (avg {t / c / 1e3:7.1f} us)   after {a}   before {b}")
