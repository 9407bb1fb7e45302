"""Prints the JSON lines of a tools/r5_tile.sh latency file as a table: python tools/tile_table.py gpurun_out/tile_latency_x.txt"""
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{") and "samples_per_tile" in l:
        d = json.loads(l)
        print("mode %d thr %2d n %5d prec %d model %d  p50 %7.2f p99 %8.2f  %6.1f Mrays/s" % (d["mode"], d["threads"], d["samples_per_tile"], d["precision"], d["lensModel"], d["p50_us"], d["p99_us"], d["mrays_s"]))
    elif l.startswith("{"):
        d = json.loads(l)
        print("per-sample thr %2d  median %.2f p99 %.2f  %.0f calls/s" % (d["threads"], d["median_us"], d["p99_us"], d["calls_per_s"]))
    else:
        print(l)
