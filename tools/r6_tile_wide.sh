#!/bin/bash
# tools/r6_tile_wide.sh -- the tile server's figures that a change of the resident kernels can move: the latency path (one thread: 4096 / 256 samples, the per-sample call),
# the throughput shapes (16 threads), large requests (65 536 samples: rows, records, device-resident)
cd $GRAFT_REPO_ROOT
L=zoic_amd/lenses/double_gauss_f2.0.dat
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: %7.1f Mrays/s  p50 %8.2f  p99 %8.2f us' % (d['mrays_s'], d['p50_us'], d['p99_us']))"; }
for rep in 1 2; do
tools/native/tile_latency $L 1 4096 2000 1 1 0 | p "1 x 4096 rows"
tools/native/tile_latency $L 1 256 2000 1 1 0 | p "1 x 256 rows"
tools/native/tile_latency $L 1 4096 2000 0 1 0 | p "1 x 4096 rows STRICT"
tools/native/tile_latency $L 16 4096 600 1 1 0 | p "16 x 4096 rows"
tools/native/tile_latency $L 1 16384 1000 1 1 0 1 1 | p "1 x 16384 records+samples16"
tools/native/tile_latency $L 1 65536 500 1 1 0 1 1 | p "1 x 65536 records+samples16"
tools/native/tile_latency $L 16 65536 60 1 1 0 1 1 | p "16 x 65536 records+samples16"
tools/native/tile_latency $L 16 65536 60 1 1 0 | p "16 x 65536 rows"
tools/native/tile_latency $L 1 65536 500 1 1 4 | p "device 1 x 65536"
tools/native/tile_latency $L 4 65536 300 1 1 4 | p "device 4 x 65536"
tools/native/tile_latency $L 1 4096 1000 1 1 4 | p "device 1 x 4096"
tools/native/sample_latency $L 1 40000 1 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-sample FAST 1 thread: median %.2f p99 %.2f us' % (d['median_us'], d['p99_us']))"
done
