TAG=r04zzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for spec in c4k:c4_klcp c4k:c4_klcp_subst1pct c4k:c4_klcp_lookup c4k:c4_klcp_lookup_fast c4k48:c4_klcp_k48; do
  leg=${spec%%:*}; v=${spec##*:}
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$v" -o run -- python3 "$R/bench.py" --leg $leg --variants $v --no-pmc > "$O/trace_$v.json" 2> "$O/trace_$v.log"
done
timeout 120 python3 "$R/tools/profile_summary3.py" "$O" > "$O/summary.log" 2>&1
find "$O" -name '*_kernel_trace.csv' -delete
find "$O" -name '*_agent_info.csv' -delete
cat "$O/kernel_stats_summary.csv" | cut -c1-200
