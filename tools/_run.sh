cd $GRAFT_REPO_ROOT
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
cp zoic_amd/libzoic_amd.so /tmp/keep.so; cp tools/ubench/libzoic_tt.so zoic_amd/libzoic_amd.so
for args in "1 4096 300 1 1 0" "1 4096 300 2 1 0"; do tools/native/tile_latency $LENS $args 2>&1 | tail -6 | cut -c1-220; done
cp /tmp/keep.so zoic_amd/libzoic_amd.so
