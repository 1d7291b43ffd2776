#!/bin/bash
# Collects the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01d
# bench lines (default workload incl. cpu_baseline, stream workload), rocprofv3 kernel trace + stats of the same bench
# command, and the PMC passes (each in its own run: FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS | SQ counters), then
# tools/pmc_summary.py condenses them into gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 10 > "$O/bench_default.json" 2> "$O/bench_default.err"
python3 "$R/bench.py" --workload stream --steps 10 > "$O/bench_stream.json" 2> "$O/bench_stream.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o run -- python3 "$R/bench.py" --steps 10 --cpu-sample 0 --no-pmc > "$O/trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_stream" -o run -- python3 "$R/bench.py" --workload stream --steps 10 --cpu-sample 0 --no-pmc > "$O/trace_stream.log" 2>&1
# (the PMC passes keep the counted launches: the LF-walk legs of the default run are part of what is profiled)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o run -- python3 "$R/bench.py" --steps 3 --cpu-sample 0 --no-pmc > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/pmc_write" -o run -- python3 "$R/bench.py" --steps 3 --cpu-sample 0 --no-lf-leg --no-pmc > "$O/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/pmc_sq" -o run -- python3 "$R/bench.py" --steps 3 --cpu-sample 0 --no-lf-leg --no-pmc > "$O/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_stream" -o run -- python3 "$R/bench.py" --workload stream --steps 3 --cpu-sample 0 --no-counters --no-pmc > "$O/pmc_fetch_stream.log" 2>&1
python3 "$R/tools/pmc_summary.py" "$O"
# the raw per-dispatch CSVs are large; the condensed files are what is kept
find "$O" -name '*_counter_collection.csv' -delete
find "$O" -name '*_kernel_trace.csv' -delete
ls -la "$O"
