#!/bin/bash
# Round 6: the `fmsi` binary on 10^8 records of one k-mer each, default mode and -O, with the prefix table's signature pass beside the first
# batches (default) and inside the load (FMSI_HIP_CLI_SYNC_TABLE=1): where does the default mode's device stage go?
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r06}
NREC=${2:-100000000}
W=/dev/shm/fmsi_exp_kmers
mkdir -p $W gpurun_out
python tools/e2e_bench.py --n 2800000000 --reads 0 --kmer-records $NREC --kmer-modes or,all,FMSI_HIP_CLI_SYNC_TABLE=1+or,FMSI_HIP_CLI_SYNC_TABLE=1+all,or,all --dir $W > gpurun_out/${TAG}_kmers_modes.txt 2>&1 || true
grep '^[a-zA-Z_=0-9+]* {' gpurun_out/${TAG}_kmers_modes.txt | python3 -c "
import sys, json
for l in sys.stdin:
    name, _, js = l.partition(' ')
    e = json.loads(js)
    print('%-36s wall %.3f load %.3f read %.3f prep %.3f device %.3f replay %.3f format %.3f write %.3f' % (name, e['wall_seconds'], e['load_seconds'], e['read_seconds'], e['prepare_seconds'], e['device_seconds'], e['replay_seconds'], e['format_seconds'], e['write_seconds']))
"
rm -rf $W
