cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM" ; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/sq_$tag -o run -- python3 $R/bench.py --pmc-child --pmc-plan '[{"name":"klcp","via":"klcp","batch":134217728},{"name":"klcp_s","via":"klcp","subst_ppm":10000,"batch":134217728},{"name":"chain","via":"runs_chain","batch":134217728},{"name":"plain","via":"runs_plain","batch":134217728}]' --workload stream --stream-via runs --prefix-len 16 --n 2.8e9 --batch 134217728 > $R/gpurun_out/sq_$tag.log 2>&1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/sq_$tag query_stream_kernel dict_chain_kernel "dict_lookup_kernel<false, true>" > $R/gpurun_out/sq_$tag.txt
  find $R/gpurun_out/sq_$tag -name '*.csv' -delete
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sq_trace -o run -- python3 $R/bench.py --pmc-child --pmc-plan '[{"name":"klcp","via":"klcp","batch":134217728},{"name":"klcp_s","via":"klcp","subst_ppm":10000,"batch":134217728},{"name":"chain","via":"runs_chain","batch":134217728},{"name":"plain","via":"runs_plain","batch":134217728}]' --workload stream --stream-via runs --prefix-len 16 --n 2.8e9 --batch 134217728 > $R/gpurun_out/sq_trace.log 2>&1
find $R/gpurun_out/sq_trace -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/sq_trace_kernel_stats.csv \;
find $R/gpurun_out/sq_trace -name '*_kernel_trace.csv' -delete
cat $R/gpurun_out/sq_*.txt
