"""Per-kernel averages of a rocprofv3 --pmc pass written with --output-format csv:
    python tools/pmc_summary.py <dir> [name substring]
(one line per kernel name: calls, then every counter's mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if needle not in name:
                continue
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row.get("Dispatch_Id"))
for name in sorted(acc, key=lambda n: -len(calls[n])):
    n = max(1, len(calls[name]))
    short = name.split("(")[0][-70:]
    print(f"{short}  calls={n}  " + "  ".join(f"{k}={v / n:.4g}" for k, v in sorted(acc[name].items())))
