#!/bin/bash
# Round 3's profile evidence, collected on the GPU box (gpurun -- bash tools/profile_round3.sh r03x):
#   * the default bench line as the driver runs it (headline + lf_path + cpu_baseline + configs legs, PMC traffic in-run);
#   * rocprofv3 --kernel-trace --stats of the same headline command and of every configs variant on its own, so that each kernel's
#     average duration can be held against the line's avg_kernel_ms;
# everything lands in gpurun_out/prof_<tag>/; tools/profile_summary3.py condenses the kernel stats.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_headline" -o run -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-pmc --no-config-legs --cpu-sample 0 > "$O/trace_headline.log" 2>&1
for spec in c2:c2 c3:c3 c4:c4_dict_chain c4:c4_dict_chain_subst1pct c4:c4_dict_plain c4:c4_dict_plain_subst1pct c4:c4_klcp c4:c4_klcp_subst1pct c4sk:c4_superkmer c4sk:c4_superkmer_subst1pct; do
  leg=${spec%%:*}; v=${spec##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$v" -o run -- python3 "$R/bench.py" --leg $leg --variants $v --no-pmc > "$O/trace_$v.json" 2> "$O/trace_$v.log"
done
python3 "$R/tools/profile_summary3.py" "$O"
find "$O" -name '*_kernel_trace.csv' -delete
find "$O" -name '*_agent_info.csv' -delete
ls -la "$O"
