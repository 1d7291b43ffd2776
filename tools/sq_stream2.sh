# SQ counters of the S-block streaming kernel (query_stream2_kernel), reads without / with 1 % substitutions (gpurun -- bash tools/sq_stream2.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PLAN='[{"name":"klcp","via":"klcp","batch":134217728},{"name":"klcp_s","via":"klcp","subst_ppm":10000,"batch":134217728}]'
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/s2_$tag -o run -- python3 $R/bench.py --pmc-child --pmc-plan "$PLAN" --workload stream --stream-via klcp --n 2.8e9 --batch 134217728 > $R/gpurun_out/s2_$tag.log 2>&1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/s2_$tag query_stream2_kernel > $R/gpurun_out/s2_$tag.txt
  find $R/gpurun_out/s2_$tag -name '*.csv' -delete
done
cat $R/gpurun_out/s2_*.txt
