#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_round.sh: per (pass, kernel, counter) the number of dispatches and
the average per dispatch for the query kernels, the kernel-stats tables, and traffic.json = HBM bytes per launch of
the timed kernel next to the algorithmic bytes bench.py counted for the same configuration.

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 tallies a 128-byte request at 64 B in FETCH_SIZE (MI355X_MICROARCH.md, HBM
section; re-measured with tools/gather_bench: 0.50x for random 128-B reads, 1.015x for random 64-B reads), so for the
pair kernel corrected fetch = raw + 64 B x (128-B block reads per launch, from the kernel's own counters)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(out_dir: str) -> None:
    rows = []
    for pass_dir in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
        if not os.path.isdir(pass_dir):
            continue
        acc = defaultdict(lambda: [0, 0.0])
        per_dispatch = defaultdict(lambda: defaultdict(float))
        for f in glob.glob(os.path.join(pass_dir, "**", "*_counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"]
                    if "query_" not in name and "pair_lookup" not in name and "lookup_lane" not in name and "dict_lookup" not in name and "dict_chain" not in name:
                        continue
                    per_dispatch[(name, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (name, counter), d in per_dispatch.items():
            acc[(name, counter)] = [len(d), sum(d.values()) / len(d)]
        for (name, counter), (cnt, avg) in sorted(acc.items()):
            rows.append((os.path.basename(pass_dir), name, counter, cnt, avg))
    with open(os.path.join(out_dir, "pmc_summary.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["pass", "kernel", "counter", "dispatches", "avg_per_dispatch"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], round(r[4], 3)])
    for tag in ("trace", "trace_stream"):
        for f in glob.glob(os.path.join(out_dir, tag, "**", "*_kernel_stats.csv"), recursive=True):
            os.replace(f, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))

    def avg(pass_name, kernel_sub, counter):
        c = [r for r in rows if r[0] == pass_name and kernel_sub in r[1] and r[2] == counter and "<false" in r[1]]
        return max(c, key=lambda r: r[3])[4] if c else None

    for wl, bench_file, kernel_sub, fetch_pass in (("kmers", "bench_default.json", "query_pair_kernel", "pmc_fetch"),
                                                   ("stream", "bench_stream.json", "query_stream_kernel", "pmc_fetch_stream")):
        try:
            line = [l for l in open(os.path.join(out_dir, bench_file)) if l.startswith("{")][-1]
            b = json.loads(line)
        except (OSError, IndexError, ValueError):
            continue
        roof = b["roofline"]
        if wl == "kmers" and roof["kernel"].startswith("dict_lookup"):
            # the k-mer dictionary: every read of the kernel is a 128-byte request (a bucket by eight lanes, 64 k-mers by a wave), which
            # gfx950 tallies at 64 B in FETCH_SIZE (MI355X_MICROARCH.md, HBM; tools/gather_bench calibration: 0.50x): fetched = 2 x raw
            raw = avg(fetch_pass, "dict_lookup_kernel", "FETCH_SIZE")
            if raw is None:
                continue
            batch = b["config"]["queries_per_step_per_gpu"]
            wr = avg("pmc_write", "dict_lookup_kernel", "WRITE_SIZE")
            t = {"kernel": "dict_lookup_kernel",
                 "config": {"workload": wl, "n": b["config"]["index_rows"] - 1, "batch": batch, "k": b["config"]["k"],
                            "prefix_len": b["config"]["prefix_table_len"], "mode": b["config"]["mode"]},
                 "fetch_size_raw_bytes_per_launch": raw * 1024.0, "fetch_bytes_per_launch": 2.0 * raw * 1024.0,
                 "write_bytes_per_launch": (wr or 0.0) * 1024.0,
                 "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
                 "tcc_hit_per_launch": avg("pmc_write", "dict_lookup_kernel", "TCC_HIT_sum"),
                 "tcc_miss_per_launch": avg("pmc_write", "dict_lookup_kernel", "TCC_MISS_sum"),
                 "correction": "2 x raw FETCH_SIZE (gfx950 tallies a 128-B request at 64 B; every read of this kernel is one)"}
            t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
            # the table-less LF-walk legs of the same run: the pair kernel reads 128-B blocks (2 x raw), the lock-step kernel 64-B
            # blocks (raw / 1.015); both re-read the top of the search tree from the caches, so traffic < algorithmic bytes
            legs = []
            for leg in (roof.get("lf_path") or []):
                sub = "query_pair_kernel<false, false, false>" if "pair" in leg["kernel"] else "query_lockstep_kernel<false>"
                lraw = avg(fetch_pass, sub, "FETCH_SIZE")
                if lraw is None:
                    continue
                corrected = 2.0 * lraw * 1024.0 if "pair" in leg["kernel"] else lraw * 1024.0 / 1.015
                legs.append({"kernel": sub, "fetch_size_raw_bytes_per_launch": lraw * 1024.0, "fetch_bytes_per_launch": corrected,
                             "algorithmic_bytes_per_launch": leg["algorithmic_bytes"], "kernel_ms": leg["ms"],
                             "hbm_gb_per_s": corrected / (leg["ms"] * 1e-3) / 1e9})
            t["lf_path"] = legs
            with open(os.path.join(out_dir, f"traffic_{wl}.json"), "w") as fh:
                json.dump(t, fh, indent=1)
                fh.write("\n")
            continue
        if wl == "stream" and roof["kernel"].startswith("dict_"):
            # the stream workload through the dictionary (chained or not): bucket reads are 128-byte requests (2 x raw); the chained
            # kernel's side reads (4..16 bytes each) are a few per cent of its traffic and are left at the counter's tally
            sub = "dict_chain_kernel" if "chain" in roof["kernel"] else "dict_lookup_kernel"
            raw = avg(fetch_pass, sub, "FETCH_SIZE")
            if raw is None:
                continue
            t = {"kernel": sub, "config": {"workload": wl, "n": b["config"]["index_rows"] - 1, "batch": b["config"]["queries_per_step_per_gpu"],
                                            "k": b["config"]["k"], "prefix_len": b["config"]["prefix_table_len"], "mode": b["config"]["mode"], "stream_via": "runs"},
                 "fetch_size_raw_bytes_per_launch": raw * 1024.0, "fetch_bytes_per_launch": 2.0 * raw * 1024.0, "hbm_bytes_per_launch": 2.0 * raw * 1024.0,
                 "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
                 "correction": "2 x raw FETCH_SIZE (128-byte bucket requests tallied at 64 B); result words not included"}
            with open(os.path.join(out_dir, f"traffic_{wl}.json"), "w") as fh:
                json.dump(t, fh, indent=1)
                fh.write("\n")
            continue
        raw = avg(fetch_pass, kernel_sub, "FETCH_SIZE")
        if raw is None:
            continue
        raw_bytes = raw * 1024.0
        # phased form: its first kernel (one lane per k-mer with the candidate tables of k - t <= 15, lane pairs otherwise)
        lookup_name = "lookup_lane_kernel" if avg(fetch_pass, "lookup_lane_kernel", "FETCH_SIZE") is not None else "pair_lookup_kernel"
        lookup_raw = avg(fetch_pass, lookup_name, "FETCH_SIZE") if wl == "kmers" else None
        batch = b["config"]["queries_per_step_per_gpu"]
        t = {"kernel": kernel_sub, "config": {"workload": wl, "n": b["config"]["index_rows"] - 1, "batch": batch, "k": b["config"]["k"],
                                                "prefix_len": b["config"]["prefix_table_len"], "mode": b["config"]["mode"]},
             "fetch_size_raw_bytes_per_launch": raw_bytes,
             "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"]}
        if wl == "kmers":
            block_reads = (roof["blocks_per_kmer"] - roof["table_lookups_per_kmer"]) * batch   # 128-B requests
            wr = avg("pmc_write", kernel_sub, "WRITE_SIZE")
            wr_lookup = avg("pmc_write", lookup_name, "WRITE_SIZE")
            t["write_bytes_per_launch"] = ((wr or 0.0) + (wr_lookup or 0.0)) * 1024.0
            t["fetch_bytes_per_launch"] = raw_bytes + 64.0 * block_reads
            if lookup_raw is not None:
                t["kernel"] = lookup_name + " + query_pair_kernel"
                t["lookup_kernel_fetch_bytes_per_launch"] = lookup_raw * 1024.0
                t["fetch_bytes_per_launch"] += lookup_raw * 1024.0
            t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
            t["tcc_hit_per_launch"] = avg("pmc_write", kernel_sub, "TCC_HIT_sum")
            t["tcc_miss_per_launch"] = avg("pmc_write", kernel_sub, "TCC_MISS_sum")
            t["correction"] = "raw FETCH_SIZE + 64 B x 128-B block reads per launch (gfx950 tallies a 128-B request at 64 B)"
        else:
            t["fetch_bytes_per_launch"] = raw_bytes / 1.015   # random 64-B reads are tallied at 1.015x (gather_bench calibration)
            t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"]
            t["correction"] = "raw FETCH_SIZE / 1.015 (64-B requests); result bytes leave through stream_compact_kernel, not this kernel"
        with open(os.path.join(out_dir, f"traffic_{wl}.json"), "w") as fh:
            json.dump(t, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main(sys.argv[1])
