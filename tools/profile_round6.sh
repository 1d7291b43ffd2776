#!/bin/bash
# Round 6's profile evidence, collected on the GPU box (gpurun -- bash tools/profile_round6.sh r06z):
#   * the default bench line as the driver runs it (full record + compact record; PMC traffic in-run; the e2e legs incl. the gzip ones);
#   * rocprofv3 --kernel-trace --stats of the headline command and of every configs variant on its own, so that each kernel's average
#     duration can be held against the line's avg_kernel_ms (tools/profile_summary3.py; the legs' one sample launch dropped).
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 700 python3 "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_headline" -o run -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-pmc --no-config-legs --cpu-sample 0 > "$O/trace_headline.log" 2>&1
for spec in c2:c2 c3sk:c3_superkmer c4k:c4_klcp c4k:c4_klcp_subst1pct c4k:c4_klcp_lookup c4k:c4_klcp_lookup_fast c4k16:c4_klcp_t16 c4k16:c4_klcp_t16_subst1pct c4k48:c4_klcp_k48 c4sk:c4_superkmer c5dict:c5_kmer_dictionary; do
  leg=${spec%%:*}; v=${spec##*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$v" -o run -- python3 "$R/bench.py" --leg $leg --variants $v --no-pmc > "$O/trace_$v.json" 2> "$O/trace_$v.log"
done
timeout 120 python3 "$R/tools/profile_summary3.py" "$O" > "$O/summary.log" 2>&1
find "$O" -name '*_kernel_trace.csv' -delete
find "$O" -name '*_agent_info.csv' -delete
tail -1 "$O/bench_default.json" | cut -c1-600
cat "$O/kernel_stats_summary.csv" | cut -c1-200
