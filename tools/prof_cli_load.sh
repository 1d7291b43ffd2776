#!/bin/bash
# Kernel trace of the `fmsi query` process's load at human scale: which kernels the 0.5 s of index load are made of.
#   gpurun -- bash tools/prof_cli_load.sh [tag]
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05}
export TMPDIR=/tmp
W=/dev/shm/fmsi_prof_load
mkdir -p $W gpurun_out
python - <<PY
import sys, numpy as np
sys.path.insert(0, ".")
import torch, fmsi_amd as fm
dev = fm.DeviceIndex.build_synthetic(2_800_000_000, 12345, 31, with_klcp=True)
dev.write_files("$W/idx")
d = torch.empty(1 << 16, dtype=torch.int64, device="cuda")
dev.synthetic_queries_dev(31, 42, 0, 1 << 16, d.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
km = d.cpu().numpy().view(np.uint64)
with open("$W/q.fa", "w") as f:
    for i, x in enumerate(km[:20000]):
        f.write(">%d\n%s\n" % (i, "".join("ACGT"[(int(x) >> (2 * t)) & 3] for t in range(31))))
PY
for mode in "" "-S"; do
  FMSI_HIP_VERBOSE=1 fmsi_amd/bin/fmsi query $mode --stats -q $W/q.fa $W/idx > /dev/null 2> gpurun_out/${TAG}_load_plain${mode}.txt || true
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_load${mode} -o load -- fmsi_amd/bin/fmsi query $mode -q $W/q.fa $W/idx > /dev/null 2> gpurun_out/${TAG}_load_rocprof${mode}.txt || true
  cp /tmp/prof_load${mode}/*kernel_stats.csv gpurun_out/${TAG}_load_kernel_stats${mode}.csv 2>/dev/null || find /tmp/prof_load${mode} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_load_kernel_stats${mode}.csv \;
  echo "== mode '$mode'"; cat gpurun_out/${TAG}_load_plain${mode}.txt; head -25 gpurun_out/${TAG}_load_kernel_stats${mode}.csv | cut -c1-200
done
rm -rf $W
du -sh gpurun_out
