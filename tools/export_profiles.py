"""Turn rocprofv3's sqlite outputs into the small summaries committed under profiles/:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python3 bench.py --no-cpu-baseline --steps 20
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o pf -- python3 bench.py --no-cpu-baseline --steps 5
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o pw -- python3 bench.py --no-cpu-baseline --steps 5
    python tools/export_profiles.py gpurun_out profiles/r01 f16x3

writes <out>/bench_kernel_stats_<tag>.csv (per-kernel calls / total / average, the --stats table) and
<out>/traffic_<tag>.json (FETCH_SIZE / WRITE_SIZE in KB per launch, averaged per kernel; bench.py applies the gfx950
x2 correction of MI355X_MICROARCH.md to FETCH_SIZE)."""
import csv, json, sqlite3, sys, os

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


c = sqlite3.connect(os.path.join(src, "prof_kt", "kt_results.db"))
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(os.path.join(out, f"bench_kernel_stats_{tag}.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
    for name, calls, total, avg, pct in rows:
        w.writerow([short(name), calls, round(total, 3), round(avg, 3), round(pct, 3)])

traffic = {"collected": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, averaged over launches (KB)", "all_kernels": {}}
for counter, sub, stem in (("FETCH_SIZE", "prof_fetch", "pf"), ("WRITE_SIZE", "prof_write", "pw")):
    c = sqlite3.connect(os.path.join(src, sub, f"{stem}_results.db"))
    q = "select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name"
    for name, val, n in c.execute(q, (counter,)):
        if "vfn_" in name:
            traffic["all_kernels"][f"{counter}|{short(name)}"] = round(val, 2)
# template arguments are the launch modes of csrc/vfn_mlp16.hip: <0> vector-only VF (grid queries), <3> fused VF + rendering,
# <9> VF with feature blocks out (once per distinct sample), <18> rendering net on gathered blocks
traffic["modes"] = {"vfn_mlp16_kernel<0>": "VF, vector columns only", "vfn_mlp16_kernel<3>": "fused VF + rendering (fine pass)",
                    "vfn_mlp16_kernel<9>": "VF + feature blocks out", "vfn_mlp16_kernel<18>": "rendering net on gathered blocks"}
with open(os.path.join(out, f"traffic_{tag}.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(open(os.path.join(out, f"traffic_{tag}.json")).read())
