#!/bin/bash
# kernel trace of the training bench -> gpurun_out/kt_<tag>.txt (top kernels)
R=$(pwd); tag=${1:-x}; shift
mkdir -p $R/gpurun_out/ktmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktmp -o kt -- python3 $R/bench.py --workload train --steps 15 --no-parity "$@" > /dev/null 2>&1 < /dev/null
cd $R
python3 - "$tag" <<'PY'
import sqlite3, sys, glob
db = glob.glob("gpurun_out/ktmp/**/kt_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = [(n, k, tot, avg) for n, k, tot, avg in c.execute("select name, total_calls, total_duration, average from top_kernels limit 14")]
with open(f"gpurun_out/kt_{sys.argv[1]}.txt", "w") as f:
    for n, k, tot, avg in rows:
        f.write(f"{k:5d} calls  avg {avg/1e3 if avg > 1e6 else avg:9.1f}  total {tot:12.1f}  {n.replace('(anonymous namespace)::', '')[:60]}\n")
PY
rm -rf gpurun_out/ktmp
