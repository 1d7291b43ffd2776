#!/usr/bin/env python3
"""Condenses tools/profile_round3.sh's rocprofv3 outputs: one CSV row per (run, query kernel) with calls and average duration, next to
the avg_kernel_ms the bench line / leg reported for the same variant (HIP events inside the library)."""
import csv
import glob
import json
import os
import re
import sys

QUERY = ("dict_lookup_kernel", "dict_chain_kernel", "sk_runs_kernel", "query_stream2_kernel", "query_stream_kernel", "stream_prep_kernel", "stream_compact_kernel",
         "pack_results_kernel", "query_pair_kernel", "query_lockstep_kernel", "sk_kmers_kernel", "sorted_walk_kernel", "sw_items_kernel", "sw_combine_kernel",
         "onesweep_iteration_kernel")


def main(out):
    rows = []
    bench = {}
    try:
        line = json.loads([l for l in open(os.path.join(out, "bench_default.json")) if l.startswith("{")][0])   # (the full record: first line)
        bench["headline"] = line["roofline"]["avg_kernel_ms"]
        for leg in line["roofline"].get("lf_path") or []:
            bench[leg["via"]] = leg["ms"]
        for c in line.get("configs", []):
            if c.get("avg_kernel_ms"):
                bench[c["name"]] = c["avg_kernel_ms"]
    except (OSError, IndexError, ValueError, KeyError):
        pass
    for d in sorted(glob.glob(os.path.join(out, "trace_*"))):
        if not os.path.isdir(d):
            continue
        run = os.path.basename(d)[len("trace_"):]
        leg_ms = None
        try:
            got = json.loads([l for l in open(d + ".json") if l.startswith("[")][-1])
            leg_ms = got[0].get("avg_kernel_ms")
        except (OSError, IndexError, ValueError):
            pass
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    name = r["Name"]
                    if not any(q in name for q in QUERY) or "<true" in name:
                        continue
                    m = re.search(r"(\w+_kernel(?:<[^(]*>)?)", name)
                    calls, avg, mn, mx = int(r["Calls"]), float(r["AverageNs"]), float(r.get("MinNs") or 0), float(r.get("MaxNs") or 0)
                    # the legs' reference check launches the same kernel ONCE on a sample of ~0.3 M windows: a launch far below the
                    # average is that one, not a timed step; and a dictionary's construction sends the whole text through the lookup
                    # kernel ONCE (a launch tens of times longer than a step) -- the timed launches' average is reported beside the raw one
                    total, n_t = calls * avg, calls
                    if n_t > 2 and mx > 5.0 * (total - mx) / (n_t - 1):
                        total, n_t = total - mx, n_t - 1
                    if n_t > 1 and 0 < mn < 0.5 * total / n_t:
                        total, n_t = total - mn, n_t - 1
                    timed_avg = total / n_t
                    rows.append({"run": run, "kernel": m.group(1) if m else name[:60], "calls": calls, "avg_ms": round(avg / 1e6, 4),
                                 "timed_launches_avg_ms": round(timed_avg / 1e6, 4),
                                 "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "under_rocprof_hip_events_avg_kernel_ms": leg_ms,
                                 "bench_line_avg_kernel_ms": bench.get(run)})
    with open(os.path.join(out, "kernel_stats_summary.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["run", "kernel", "calls", "avg_ms", "timed_launches_avg_ms", "total_ms", "under_rocprof_hip_events_avg_kernel_ms", "bench_line_avg_kernel_ms"])
        w.writeheader()
        for r in rows:
            w.writerow(r)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main(sys.argv[1])
