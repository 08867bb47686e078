#!/bin/bash
# per-kernel MEDIAN / min / max launch duration of a bench.py run (an average hides cold first launches) -> gpurun_out/km_<tag>.txt
#   bash tools/kernel_median.sh <tag> [bench.py arguments; default: --workload train --steps 15 --no-parity]
R=$(pwd); tag=${1:-x}; shift
mkdir -p $R/gpurun_out/ktmp
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- --workload train --steps 15 --no-parity; fi
rocprofv3 --kernel-trace -d $R/gpurun_out/ktmp -o kt -- python3 $R/bench.py "$@" > /dev/null 2>&1 < /dev/null
cd $R
python3 - "$tag" <<'PY'
import sqlite3, sys, glob, statistics
db = glob.glob("gpurun_out/ktmp/**/kt_results.db", recursive=True)[0]
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in names if t.startswith("rocpd_kernel_dispatch")]
ks = [t for t in names if t.startswith("rocpd_info_kernel_symbol")]
out = open(f"gpurun_out/km_{sys.argv[1]}.txt", "w")
if not kd or not ks:
    out.write("tables: " + " ".join(names) + "\n")
else:
    cols = [r[1] for r in c.execute(f"pragma table_info({kd[0]})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks[0]})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    rows = c.execute(f"select s.{name_col}, d.end - d.start from {kd[0]} d join {ks[0]} s on d.kernel_id = s.id").fetchall()
    by = {}
    for n, dur in rows:
        by.setdefault(n, []).append(dur / 1e3)
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:24]:
        out.write(f"{len(v):5d} calls  median {statistics.median(v):9.1f}  min {min(v):9.1f}  max {max(v):9.1f}  mean {sum(v)/len(v):9.1f} us  {n.replace('(anonymous namespace)::','')[:60]}\n")
PY
rm -rf gpurun_out/ktmp
