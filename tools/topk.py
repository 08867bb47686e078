"""Print the --stats table of a rocprofv3 sqlite output (durations are stored in microseconds... as observed on this
image: `top_kernels.average` of a 0.5 ms kernel reads 500): python tools/topk.py <results.db> [n]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{pct:6.2f}%  calls {calls:6d}  avg {avg:10.1f} us  total {total / 1e3:9.2f} ms  {short[:90]}")
