cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5p32; mkdir -p $O
MISSLAP_TILED_P32=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-cpu > $O/under_rocprof.json 2> $O/stats.err || { tail -3 $O/stats.err; exit 1; }
grep -E "k_bid_tiled|k_bid_undecided|k_tile_mirror|fillBuffer" $O/stats/*kernel_stats.csv | cut -c1-200
rm -rf $O/stats
