#!/bin/bash
# Round 5, beside tools/e2e_bench.py: the same query files through a pipe and as ONE gzip member (the stream-stretch reader), and short reads.
#   gpurun -- bash tools/e2e_extra.sh
cd "$(dirname "$0")/.."
W=/dev/shm/fmsi_e2e
FMSI=fmsi_amd/bin/fmsi
python tools/e2e_bench.py --n 2800000000 --reads 100000 --modes or,all,all_S --kmer-records 100000000 --kmer-modes or,all,all_S --keep --out gpurun_out/r05i_e2e.json 2>&1 | grep -v '^{"n"' | cut -c1-260
t() { local s=$(date +%s.%N); "$@"; local e=$(date +%s.%N); echo "    wall $(echo "$e - $s" | bc) s"; }
echo "== 10^8 k-mer records through a pipe (cat | fmsi query -q -)"
t bash -c "cat $W/kmers.fa | $FMSI query --stats -q - $W/idx > $W/o1.txt" 2>&1 | cut -c1-400
echo "== the same, general path (FMSI_HIP_NO_STREAM_STRETCHES=1: round 4's ordinary reader)"
head -n 40000000 $W/kmers.fa > $W/k20.fa
t bash -c "cat $W/k20.fa | FMSI_HIP_NO_STREAM_STRETCHES=1 $FMSI query --stats -q - $W/idx > $W/o2.txt" 2>&1 | cut -c1-400
echo "== 2 x 10^7 k-mer records as one gzip member"
gzip -1 -k $W/k20.fa
t $FMSI query --stats -q $W/k20.fa.gz $W/idx > $W/o3.txt 2> $W/e3.txt; cut -c1-400 $W/e3.txt
FMSI_HIP_NO_STREAM_STRETCHES=1 $FMSI query --stats -q $W/k20.fa.gz $W/idx 2>&1 > $W/o4.txt | cut -c1-400
cmp $W/o3.txt $W/o4.txt && cmp $W/o2.txt $W/o3.txt && echo "    outputs identical"
echo "== 10-kb reads through a pipe"
t bash -c "cat $W/reads.fa | $FMSI query --stats -q - $W/idx > $W/o5.txt" 2>&1 | cut -c1-400
t bash -c "cat $W/reads.fa | FMSI_HIP_NO_STREAM_STRETCHES=1 $FMSI query --stats -q - $W/idx > $W/o6.txt" 2>&1 | cut -c1-400
cmp $W/o5.txt $W/o6.txt && echo "    outputs identical"
rm -rf $W
echo "== short reads: 6 x 10^6 reads of 150 bases"
python tools/e2e_bench.py --n 2800000000 --reads 6000000 --read-len 150 --modes or,all_S --ref-reads 2000 --out gpurun_out/r05i_e2e_short_reads.json 2>&1 | grep -v '^{"n"' | cut -c1-700
