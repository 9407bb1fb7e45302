export TMPDIR=/tmp
for m in fast unchecked strict; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_c2_$m -- python tools/kbench.py --configs C2 --modes $m > /dev/null 2>&1
python - $m <<PY
import csv,glob,sys
for f in glob.glob("gpurun_out/trace_c2_%s/*/*kernel_stats.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if "kolb" in r["Name"] or "fill" in r["Name"]: print(sys.argv[1], r["Name"][:60], r["Calls"], "%.1f" % (float(r["AverageNs"])/1e3))
PY
done
ZOIC_DEBUG_LISTS=1 python tools/kbench.py --configs C2 --modes fast --steps 1 2>&1 | grep "handed" | tail -1
