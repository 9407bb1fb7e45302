cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/bench_r05_a.json 2> gpurun_out/bench_r05_a.err; tail -c 400 gpurun_out/bench_r05_a.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r05_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
print(d.get('parity'))
print(json.dumps(d['host_path'], indent=0)[:3000])
for c in d.get('configs',[]): print(c['config'], c['mode'], c['value'], c.get('bound'), c.get('binding_frac'), c.get('flips'), c.get('rmse'))
print(d.get('frame_api'))
print(d.get('cpu_baseline'))
PY
