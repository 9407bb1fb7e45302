cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2c
( time python bench.py > gpurun_out/r2c/bench_default.json 2> gpurun_out/r2c/bench_default.err ) 2> gpurun_out/r2c/time.txt; tail -3 gpurun_out/r2c/time.txt; tail -c 400 gpurun_out/r2c/bench_default.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r2c/bench_default.json").read().strip().splitlines()[-1])
print("headline", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l.get("parity"))
for e in l.get("configs", []): print(e["config"], e["precision_mode"], e["value"], e["roofline"]["kernel_ms"], e.get("parity", {}).get("decision_flip_frac"), e.get("parity", {}).get("dir_rmse"))
for e in l.get("sharded_frame", []): print("sharded", e["config"], e["compute_only"])
print("host", l.get("host_path")); print("cpu", {k: v for k, v in l.get("cpu_baseline", {}).items() if k in ("value", "one_thread_value", "cores")})
PY
bash tools/profile_all.sh r02
