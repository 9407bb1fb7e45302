cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; echo rc=$?; tail -c 200 gpurun_out/bench_r05.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:d['roofline'][k] for k in ('bound','achieved','frac','traffic','valu_frac','lane_instr','hbm_frac')})
print(d['host_path']['tile']); print(d['host_path']['per_sample'])
for c in d.get('configs',[]): print(c['config'], c['mode'], c['value'], c.get('bound'), c.get('binding_frac'), c.get('traffic'))
print(d['frame_api']); print(d['cpu_baseline']['value'], len(open('gpurun_out/bench_r05.json').read()))
PY
