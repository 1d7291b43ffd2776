cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PLAN='[{"name":"klcp_s","via":"klcp","subst_ppm":10000,"batch":134217728}]'
timeout 50 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/s4 -o run -- python3 $R/bench.py --pmc-child --pmc-plan "$PLAN" --workload stream --stream-via klcp --n 2.8e9 --batch 134217728 --prefix-len 16 --accel tables > $R/gpurun_out/s4.log 2>&1
python3 $R/tools/pmc_quick.py $R/gpurun_out/s4 query_stream2_kernel > $R/gpurun_out/s4.txt
find $R/gpurun_out/s4 -name '*.csv' -delete
cat $R/gpurun_out/s4.txt
