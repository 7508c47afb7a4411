#!/bin/bash
# rocprofv3 PMC pass over the kernel micro-benchmarks:  tools/pmc_kernels.sh <outdir> "<counters>" <bench args...>
set -u
R=$PWD
OUT=$R/gpurun_out/$1; shift
PMC="$1"; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/bench_kernels.py "$@" --iters 2 > $OUT/stdout.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/*counter_collection.csv")
if not f:
    print("no counter csv", glob.glob("$OUT/*")); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
seen=set()
for r in rows:
    key=(r["Kernel_Name"][:60], r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[r["Kernel_Name"][:60]] += 1
for k, v in agg.items():
    if "im360" in k:
        print(k, "dispatches", cnt[k], {a: f"{b / cnt[k]:.3g}" for a, b in v.items()})
PY
rm -f $OUT/*kernel_trace.csv
