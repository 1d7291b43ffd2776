cd $GRAFT_REPO_ROOT
C="--gpus 1 --steps 20 --warmup 3 --no-pmc --no-lf-leg --no-config-legs --cpu-sample 0"
python bench.py $C > gpurun_out/cmp_plain.json 2>/dev/null
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29549 bench.py $C --force-dist"
$T > gpurun_out/cmp_dist.json 2>/dev/null

python bench.py $C --force-dist > gpurun_out/cmp_forcedist_notorchrun.json 2>/dev/null
python - <<'PY'
import json
for f in ("cmp_plain","cmp_dist","cmp_forcedist_notorchrun"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        g=d["config"].get("gather") or {}
        print(f, round(d["value"]/1e9,2), "ms/step", round(d["ms_per_step"],3), "kernel", d["roofline"]["avg_kernel_ms"], "enqueue", d["config"]["host_enqueue_ms_per_step"], g.get("gather_ms_per_step_rank0"), g.get("d2h_ms_per_step_rank0"))
    except Exception as e:
        print(f, "failed", e)
PY
