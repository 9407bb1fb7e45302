cd $GRAFT_REPO_ROOT
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
timeout -k 10 900 python -m pytest tests/test_tile_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | tail -4
for args in "1 16 1000 1 1 0" "1 256 1000 1 1 0" "1 1024 1000 1 1 0" "1 4096 1000 1 1 0" "1 4096 1000 0 1 0" "1 8192 500 1 1 0" "1 16384 500 1 1 0" "1 65536 200 1 1 0" "16 4096 500 1 1 0" "16 65536 60 1 1 0" "1 4096 1000 1 0 0"; do tools/native/tile_latency $LENS $args 2>&1 | tail -1 | cut -c1-200; done | tee gpurun_out/tile_latency_v6.txt
for t in tessar_f2.8 fisheye_muller_f4.0 petzval_f1.25; do echo $t; timeout -k 5 120 tools/native/tile_latency zoic_amd/lenses/$t.dat 1 4096 1000 1 1 0 | tail -1| cut -c1-200; done | tee -a gpurun_out/tile_latency_v6.txt
