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

import hashlib
_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# bench.py compares this fingerprint of the dominant kernel's source (csrc/vfn_mlp16.hip) with the tree it runs in and says so when
# the counters are older than the code
_src = b"".join(open(os.path.join(_repo, "vf_nerf_amd", "csrc", f), "rb").read() for f in ("vfn_mlp16.hip",))
traffic = {"collected": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, averaged over launches (KB)",
           "kernel_sources_sha16": hashlib.sha256(_src).hexdigest()[:16], "all_kernels": {}}
for counter, sub, stem in (("FETCH_SIZE", "prof_fetch", "pf"), ("WRITE_SIZE", "prof_write", "pw")):
    c = sqlite3.connect(os.path.join(src, sub, f"{stem}_results.db"))
    q = "select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name"
    for name, val, n in c.execute(q, (counter,)):
        if "vfn_" in name:
            traffic["all_kernels"][f"{counter}|{short(name)}"] = round(val, 2)
# template arguments are the launch modes of csrc/vfn_mlp16.hip: <0> vector-only VF (grid queries), <3> fused VF + rendering,
# <9> VF with feature blocks out (once per distinct sample), <18> rendering net on gathered blocks
traffic["modes"] = {"vfn_mlp16_kernel<0>": "VF, vector columns only", "vfn_mlp16_kernel<3>": "fused VF + rendering, three products everywhere",
                    "vfn_mlp16_kernel<35>": "fused VF + rendering, colour branch on two products (the default render; <3> then only appears as the guard's 128-ray self-check)",
                    "vfn_mlp16_kernel<9>": "VF + feature blocks out", "vfn_mlp16_kernel<18>": "rendering net on gathered blocks"}
with open(os.path.join(out, f"traffic_{tag}.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(open(os.path.join(out, f"traffic_{tag}.json")).read())

# optional: matrix-pipe utilisation counters (a separate pass: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
# GRBM_GUI_ACTIVE -d <src>/pmc_mfma -o pm -- python3 bench.py ...).  Per kernel: the counters averaged over launches, the launch
# duration from the kernel trace, and the derived figures of MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 / duration,
# matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x clock cycles of the launch).
pm = os.path.join(src, "pmc_mfma", "pm_results.db")
if os.path.exists(pm):
    c = sqlite3.connect(pm)
    per = {}
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
    for name, counter, val, n in c.execute(q):
        if "vfn_" in name:
            per.setdefault(short(name), {})[counter] = round(val, 1)
            per[short(name)]["launches"] = n
    dur = {short(name): avg for name, calls, total, avg, pct in rows}
    for k, v in per.items():
        us = dur.get(k)
        if us and "GRBM_GUI_ACTIVE" in v:
            clock_ghz = v["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)
            v["avg_launch_us_kernel_trace"] = round(us, 2)
            v["effective_clock_ghz"] = round(clock_ghz, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                v["mfma_busy_fraction_of_1024_simds"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clock_ghz * us * 1e3), 4)
    with open(os.path.join(out, f"pmc_mfma_{tag}.json"), "w") as fh:
        json.dump({"collected": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass), averaged over "
                                "launches; profiled passes run at a lower clock than un-profiled ones (MI355X_MICROARCH.md, DVFS (2))",
                   "kernels": per}, fh, indent=1)
    print(open(os.path.join(out, f"pmc_mfma_{tag}.json")).read())
