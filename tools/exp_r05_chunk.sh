# kLCP kernel: tasks per wave (FMSI_HIP_STREAM_CHUNK) against the read mix, t = 15 (gpurun -- bash tools/exp_r05_chunk.sh)
R=${GRAFT_REPO_ROOT:-.}
for sub in 0 10000; do
  for chunk in 64 128 256 1024; do
    FMSI_HIP_STREAM_CHUNK=$chunk timeout 300 python3 $R/bench.py --workload stream --stream-via klcp --n 2.8e9 --batch 134217728 --prefix-len 15 --accel tables \
      --subst-ppm $sub --steps 8 --warmup 2 --no-config-legs --no-lf-leg --no-pmc --cpu-sample 0 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('subst $sub chunk $chunk', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'))"
  done
done
