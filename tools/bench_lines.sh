#!/bin/bash
# The round's bench lines beyond the default one (run through gpurun from the repo root): tools/bench_lines.sh r01k
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lines_$TAG.jsonl
: > "$O"
run() { timeout 400 python3 "$R/bench.py" "$@" 2>> "$R/gpurun_out/lines_$TAG.err" < /dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['_command']='python bench.py $*'; print(json.dumps(d))" >> "$O"; }
run --steps 10 --abi-gather
run --no-pmc --workload stream --steps 10
run --no-pmc --workload stream --subst-ppm 10000 --steps 10 --cpu-sample 0
run --no-pmc --workload stream --no-dict-chain --steps 10 --cpu-sample 0
run --no-pmc --workload stream --no-dict-chain --subst-ppm 10000 --steps 10 --cpu-sample 0
run --no-pmc --workload stream --stream-via klcp --steps 10 --cpu-sample 0
run --no-pmc --workload stream --stream-via klcp --subst-ppm 10000 --steps 10 --cpu-sample 0
run --no-pmc --repeat-permille 300 --steps 10 --cpu-sample 0 --no-lf-leg
run --no-pmc --repeat-permille 300 --repeat-divergence-permille 20 --steps 10 --no-lf-leg
run --no-pmc --n 4.7e6 --steps 10
run --no-pmc --n 1.4e9 --steps 10
run --no-pmc --dict-sparse --steps 10 --cpu-sample 0 --no-lf-leg
run --no-pmc --accel tables --steps 10 --cpu-sample 0 --no-lf-leg
run --no-pmc --hbm-budget 8 --steps 10 --cpu-sample 0 --no-lf-leg
run --no-pmc --hbm-budget 16 --steps 10 --cpu-sample 0 --no-lf-leg
wc -l "$O"
