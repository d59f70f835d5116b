cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT MISSLAP_LIB=$GRAFT_REPO_ROOT/sslap_amd/libmisslap_diag.so
O=$GRAFT_REPO_ROOT/gpurun_out/r5abl; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/diag.py C3 --tiled-ablate > $O/diag.json 2> $O/diag.err || { tail -5 $O/diag.err; exit 1; }
python3 - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_bid_tiled" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"][:90]
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    print(n, round(v / n * 64 / 1e6 * 2, 1), "MB/launch (FETCH_SIZE x 64 B x 2)", k)
PY
rm -rf $O/pmc
