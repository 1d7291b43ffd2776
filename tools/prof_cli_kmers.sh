#!/bin/bash
# Which kernels answer a query file of ONE k-mer per record through the `fmsi` binary, default mode against -O, at human scale (round 6: the
# default mode's device stage was 0.20 s per 10^8 records where -O's -- both strands of every k-mer -- was 0.12 s).
#   gpurun -- bash tools/prof_cli_kmers.sh [tag] [records]
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r06}
NREC=${2:-30000000}
export TMPDIR=/tmp
W=/dev/shm/fmsi_prof_kmers
mkdir -p $W gpurun_out
python tools/e2e_bench.py --n 2800000000 --reads 0 --kmer-records $NREC --kmer-modes or,all --dir $W --keep > gpurun_out/${TAG}_kmers_e2e.txt 2>&1 || true
grep -o '^[a-z_]* {"rc": 0, "wall_seconds": [0-9.]*\|"device_seconds": [0-9.]*' gpurun_out/${TAG}_kmers_e2e.txt | tr '\n' ' '; echo
for mode in "" "-O"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kmers${mode} -o run -- fmsi_amd/bin/fmsi query $mode -q $W/kmers.fa $W/idx > /dev/null 2> gpurun_out/${TAG}_kmers_rocprof${mode}.txt || true
  find /tmp/prof_kmers${mode} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kmers_kernel_stats${mode}.csv \;
  echo "== mode '$mode'"; head -14 gpurun_out/${TAG}_kmers_kernel_stats${mode}.csv | cut -c1-220
done
rm -rf $W
