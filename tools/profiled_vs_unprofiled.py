#!/usr/bin/env python3
"""The dominant kernel's launch duration three ways, from ONE box in ONE lease (tools/profile_round.sh runs the two commands back to back):

  (1) the un-profiled bench line: HIP events around the fused launches inside the timed region (`roofline.avg_launch_ms`) and the shader
      clock those launches ran at (`roofline.effective_clock_ghz`: in-kernel s_memtime / s_memrealtime stamps of every workgroup);
  (2) the SAME command under `rocprofv3 --kernel-trace --stats`: the bench line that run prints (its own HIP events and in-kernel clock), and
  (3) the profiler's kernel-trace durations of that run: the --stats average over ALL launches (run-in included) and the average over the
      launches of the timed steps only (the last 2 x steps launches of the kernel).

A launch takes cycles / clock.  The table prints cycles = duration x in-kernel clock for (1) and (2): equal cycle counts mean the code ran
the same and the difference between a profiled and an un-profiled duration is the chip's clock under the profiler, not the kernel.

    python tools/profiled_vs_unprofiled.py <unprofiled_line.json> <profiled_run.log> <kt_results.db> > profiles/rNN/profiled_vs_unprofiled.md
"""
import json
import sqlite3
import sys


def line_of(path):
    last = None
    for raw in open(path, errors="replace"):
        raw = raw.strip()
        if raw.startswith('{"metric"'):
            last = json.loads(raw)
    if last is None:
        sys.exit(f"no bench line in {path}")
    return last


def main():
    un, pr = line_of(sys.argv[1]), line_of(sys.argv[2])
    db = sqlite3.connect(sys.argv[3])
    sym = "vfn_mlp16_kernel<3>"
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    durs = [(e - s) / 1e6 for n, s, e in db.execute(f"select {name_col}, start, end from {view} order by start")
            if n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") == sym]
    steps = int(pr["steps"])
    timed = durs[-2 * steps:]
    stats_avg = None
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") == sym:
            stats_avg = (avg / 1e3, calls)
    ru, rp = un["roofline"], pr["roofline"]

    def cyc(r):
        return r["avg_launch_ms"] * 1e-3 * r["effective_clock_ghz"] * 1e9 if r.get("effective_clock_ghz") else None
    print(f"# `{sym}` — un-profiled line vs `rocprofv3 --kernel-trace --stats`, same box, same lease, back to back\n")
    print(f"scene: {un['weights']['trained_by']}; {un['config']['rays_per_chunk_per_gpu']} rays x {un['config']['samples_per_ray']} samples, {steps} timed steps\n")
    print("| run | how the duration is measured | avg launch (ms) | in-kernel clock (GHz) | cycles per launch | roofline frac |")
    print("|---|---|---|---|---|---|")
    print(f"| un-profiled | HIP events in the timed region (the line's `roofline.avg_launch_ms`) | {ru['avg_launch_ms']:.4f} | {ru['effective_clock_ghz']} | "
          f"{cyc(ru):.4g} | {ru['frac']} |")
    print(f"| under rocprofv3 | HIP events in the timed region (the profiled run's own line) | {rp['avg_launch_ms']:.4f} | {rp['effective_clock_ghz']} | "
          f"{cyc(rp):.4g} | {rp['frac']} |")
    if timed:
        t_avg = sum(timed) / len(timed)
        flops = rp["flops_per_launch"]
        print(f"| under rocprofv3 | kernel trace, the {len(timed)} launches of the timed steps | {t_avg:.4f} | (same run) | "
              f"{t_avg * 1e-3 * rp['effective_clock_ghz'] * 1e9:.4g} | {flops / (t_avg * 1e-3) / 1e12 / rp['peak']:.4f} |")
    if stats_avg:
        print(f"| under rocprofv3 | kernel trace, `--stats` average over all {stats_avg[1]} launches (run-in included) | {stats_avg[0]:.4f} | | | "
              f"{rp['flops_per_launch'] / (stats_avg[0] * 1e-3) / 1e12 / rp['peak']:.4f} |")
    print()
    ev_vs_trace = (sum(timed) / len(timed)) / rp["avg_launch_ms"] if timed else None
    print(f"* the profiled run's HIP events and the profiler's own timestamps of the same launches agree to {abs(ev_vs_trace - 1) * 100:.1f} % "
          f"(trace / events = {ev_vs_trace:.4f}): the line's event timing measures what the profiler measures.")
    print(f"* profiled / un-profiled duration = {rp['avg_launch_ms'] / ru['avg_launch_ms']:.4f}; profiled / un-profiled clock = "
          f"{rp['effective_clock_ghz'] / ru['effective_clock_ghz']:.4f}; cycles per launch profiled / un-profiled = {cyc(rp) / cyc(ru):.4f} "
          f"(1.00 = the same code at a different clock).")
    print(f"* `value`: un-profiled {un['value']:.0f} rays/s ({un['ms_per_step']} ms per step), profiled {pr['value']:.0f} rays/s ({pr['ms_per_step']} ms).")


if __name__ == "__main__":
    main()
