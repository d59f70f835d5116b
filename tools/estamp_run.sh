# diagnostic: epilogue / loop cycles of k_bid_tiled, partial rounds (lib_estamp3) and full scans (lib_estamp4), shapes 0 and 4
mkdir -p gpurun_out/fs4
for v in 3 4; do for sh in 0 4; do
  MISSLAP_LIB=$PWD/build_ab/lib_estamp$v.so timeout -k 10 200 python tools/tail_stats.py C3 1 tiled_shape=$sh > gpurun_out/fs4/s${v}_shape$sh.json 2> gpurun_out/fs4/s${v}_shape$sh.err || { echo fail; exit 1; }
done; done; echo done
