# SQ counters of the kLCP kernel at the CLI's table depth (t = 15), clean mix and 1 % substitutions, one wave per workgroup (round 5) and four
# (FMSI_HIP_STREAM_WPB=4: rounds 3-4) -- gpurun -- bash tools/sq_stream2_r05.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wpb in 1 4; do
  for sub in 0 10000; do
    PLAN="[{\"name\":\"klcp\",\"via\":\"klcp\",\"subst_ppm\":$sub,\"batch\":134217728}]"
    for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
      tag=w${wpb}_s${sub}_$(echo $pass | cut -d' ' -f1)
      FMSI_HIP_STREAM_WPB=$wpb timeout 55 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/s5_$tag -o run -- python3 $R/bench.py --pmc-child --pmc-plan "$PLAN" --workload stream --stream-via klcp --n 2.8e9 --batch 134217728 --prefix-len 15 --accel tables > $R/gpurun_out/s5_$tag.log 2>&1
      echo "== waves per workgroup $wpb, subst_ppm $sub: $pass" >> $R/gpurun_out/s5_summary.txt
      python3 $R/tools/pmc_quick.py $R/gpurun_out/s5_$tag query_stream2_kernel >> $R/gpurun_out/s5_summary.txt
      find $R/gpurun_out/s5_$tag -name '*.csv' -delete
    done
  done
done
cat $R/gpurun_out/s5_summary.txt
