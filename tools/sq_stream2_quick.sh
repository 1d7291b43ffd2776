# SQ counters of the final S-block kLCP kernel on the benchmark's clean read mix, two passes (gpurun -- bash tools/sq_stream2_quick.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PLAN='[{"name":"klcp","via":"klcp","batch":134217728}]'
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 55 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/s3_$tag -o run -- python3 $R/bench.py --pmc-child --pmc-plan "$PLAN" --workload stream --stream-via klcp --n 2.8e9 --batch 134217728 --prefix-len 16 --accel tables > $R/gpurun_out/s3_$tag.log 2>&1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/s3_$tag query_stream2_kernel > $R/gpurun_out/s3_$tag.txt
  find $R/gpurun_out/s3_$tag -name '*.csv' -delete
done
cat $R/gpurun_out/s3_*.txt
