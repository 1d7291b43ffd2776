# SQ counters + kernel stats of the super-k-mer dictionary's kernel on the c4sk leg's handle (run on the GPU box: gpurun -- bash tools/sq_sk.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PLAN='[{"name":"sk","via":"runs_sk","batch":134217728},{"name":"sk_s","via":"runs_sk","subst_ppm":10000,"batch":134217728}]'
ARGS="--pmc-child --workload stream --stream-via runs --n 2.8e9 --batch 134217728"
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/sk_$tag -o run -- python3 $R/bench.py $ARGS --pmc-plan "$PLAN" > $R/gpurun_out/sk_$tag.log 2>&1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/sk_$tag sk_runs_kernel > $R/gpurun_out/sk_$tag.txt
  find $R/gpurun_out/sk_$tag -name '*.csv' -delete
done
cat $R/gpurun_out/sk_*.txt
