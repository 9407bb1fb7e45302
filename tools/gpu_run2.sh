cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2b
( time timeout 900 python bench.py > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err ) 2> gpurun_out/r2b/bench_default.time
tail -c 1500 gpurun_out/r2b/bench_default.err; cat gpurun_out/r2b/bench_default.time
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r2b/bench_default.json").read().strip().splitlines()[-1])
print("headline", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l.get("parity"))
for e in l.get("configs", []): print(e["config"], e["precision_mode"], e["value"], e["roofline"]["kernel_ms"], e.get("parity", {}).get("decision_flip_frac"), e.get("parity", {}).get("dir_rmse"))
for e in l.get("sharded_frame", []): print("sharded", e["config"], e["compute_only"], e["parallelism"])
print("host", l.get("host_path")); print("cpu", l.get("cpu_baseline"))
PY
ZOIC_FORCE_DIST=1 NCCL_DEBUG=INFO timeout 300 python bench.py --no-configs --no-cpu-baseline --no-parity --no-host-path > gpurun_out/r2b/bench_forcedist.json 2> gpurun_out/r2b/rccl_force_dist.log; tail -c 300 gpurun_out/r2b/bench_forcedist.json; grep -c NCCL gpurun_out/r2b/rccl_force_dist.log
timeout 600 python -m pytest tests/test_sharding_cpu.py -m gpu -x -q 2>&1 | tail -3
