#!/usr/bin/env python3
"""One training step's kernels on a time axis, from a rocprofv3 --kernel-trace database: start offset, duration, queue, name — for the LAST
complete step of the run (a step starts at the kernel whose name contains `first`, default the step's prep kernel).
python tools/trace_timeline.py <results.db> [first_kernel_substring] [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "vfn_train_prep"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else "kernel_name"
q_col = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = list(db.execute(f"select {name_col}, start, end{', ' + q_col if q_col else ''} from {view} order by start"))
starts = [i for i, r in enumerate(rows) if first in r[0]]
if len(starts) < 3:
    sys.exit(f"fewer than three kernels named *{first}* in the trace")
lo, hi = starts[-2], starts[-1]
t0 = rows[lo][1]
print(f"step of {hi - lo} kernels, {(rows[hi][1] - t0) / 1e3:.1f} us from its first kernel to the next step's first")
queues = {}
for r in rows[lo:hi]:
    name = r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44]
    q = queues.setdefault(r[3], len(queues)) if q_col else 0
    if (r[2] - r[1]) / 1e3 >= min_us:
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f}  {(r[2] - r[1]) / 1e3:8.1f} us  q{q}  {name}")
