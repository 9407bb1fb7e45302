#!/bin/bash
# tools/r6_tile_threads.sh -- more render threads than the box's CPU quota: 4096-sample tiles from 16 / 64 / 128 threads in the three wait modes
# (zoic_camera_set_wait_mode: 0 spin, 1 yield, 2 sleep); columns of tools/native/tile_latency's JSON line.
cd $GRAFT_REPO_ROOT
L=zoic_amd/lenses/double_gauss_f2.0.dat
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for threads in 16 64 128; do for wait in 0 1 2; do
  tiles=$((9600 / threads))
  tools/native/tile_latency $L $threads 4096 $tiles 1 1 0 0 0 $wait | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads %3d wait %d: %7.1f Mrays/s  p50 %8.1f  p90 %8.1f  p99 %9.1f us' % (d['threads'], d['wait_mode'], d['mrays_s'], d['p50_us'], d['p90_us'], d['p99_us']))"
done; done
for threads in 64 128; do for wait in 0 2; do
  tools/native/tile_latency $L $threads 65536 20 1 1 0 1 1 $wait | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('65536-sample tiles (records, samples16) threads %3d wait %d: %7.1f Mrays/s  p50 %8.1f  p99 %9.1f us' % (d['threads'], d['wait_mode'], d['mrays_s'], d['p50_us'], d['p99_us']))"
done; done
