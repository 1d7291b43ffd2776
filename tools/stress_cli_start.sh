#!/bin/bash
# Start-up stress of `fmsi query` on a k-mer list: N runs, exit codes counted (a crash in the first milliseconds showed once in round 5).
#   gpurun -- bash tools/stress_cli_start.sh [runs]
cd "$(dirname "$0")/.."
N=${1:-60}
python tools/e2e_bench.py --n 100000000 --reads 0 --kmer-records 5000000 --kmer-modes or --keep > /dev/null 2>&1
W=/dev/shm/fmsi_e2e
ok=0; bad=0
for i in $(seq $N); do
  fmsi_amd/bin/fmsi query -q $W/kmers.fa $W/idx > $W/o.txt 2> $W/e.txt; rc=$?
  if [ $rc -eq 0 ] && [ $(wc -l < $W/o.txt) -eq 5000000 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i: rc=$rc lines=$(wc -l < $W/o.txt) $(head -c 200 $W/e.txt)"; fi
done
echo "ok=$ok bad=$bad"
for i in $(seq 20); do
  FMSI_HIP_CLI_KMER_RANGE=4 fmsi_amd/bin/fmsi query -O -q $W/kmers.fa $W/idx > $W/o.txt 2> $W/e.txt; rc=$?
  [ $rc -eq 0 ] || echo "range4 run $i: rc=$rc"
done
echo "range4 done"
rm -rf $W
