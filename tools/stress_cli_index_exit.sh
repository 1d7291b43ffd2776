#!/bin/bash
# Round 6: `fmsi index` on a small superstring, N times (with a second binary fmsi_amd/bin/fmsi_before_exit_fix beside it, if there is one): how
# often does the process end with a non-zero status after "Written index"?  The fuzzer met ONE such run among 2 000 seeds (about 620 `fmsi index`
# runs); 2 x 1 500 runs of this script and 25 repeats of that seed met none (profiles/r06k_*).
#   gpurun -- bash tools/stress_cli_index_exit.sh [N]
cd "$(dirname "$0")/.."
N=${1:-1500}
W=/dev/shm/fmsi_stress_exit
mkdir -p $W
python3 - <<PY
import random
rng = random.Random(5)
k = 31
s = "".join(rng.choice("ACGT") for _ in range(1671))
open("$W/ms.fa", "w").write(">ms\n" + s[:-(k-1)] + s[-(k-1):].lower() + "\n")
PY
for bin in fmsi_amd/bin/fmsi_before_exit_fix fmsi_amd/bin/fmsi; do
  [ -x $bin ] || continue
  bad=0; codes=""
  s=$(date +%s.%N)
  for i in $(seq $N); do
    $bin index -k 31 $W/ms.fa > /dev/null 2> $W/err.txt
    rc=$?
    if [ $rc -ne 0 ]; then bad=$((bad+1)); codes="$codes $rc"; tail -2 $W/err.txt | tr '\n' '|' >> $W/bad_$(basename $bin).txt; echo " rc=$rc" >> $W/bad_$(basename $bin).txt; fi
  done
  e=$(date +%s.%N)
  echo "$bin: $N runs, $bad non-zero exits ($codes), $(python3 -c "print(round(($e-$s)/$N*1000,1))") ms per run"
  [ -f $W/bad_$(basename $bin).txt ] && head -5 $W/bad_$(basename $bin).txt
done
rm -rf $W
