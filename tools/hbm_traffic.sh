#!/bin/bash
# HBM traffic of the conv kernel at the cfg2 convolution shapes (tools/bench_kernels.py conv; rocprofv3 PMC, two
# separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Do NOT point this at bench.py: under --pmc the model build alone (tens of thousands of tiny init kernels) takes
# tens of minutes.   tools/hbm_traffic.sh <out_dir_under_gpurun_out>
set -u
R=$PWD
OUT=$R/gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- \
    python $R/tools/bench_kernels.py conv --iters 2 > $OUT/$C.log 2>&1
  find $OUT/$C -name "*kernel_trace.csv" -delete
done
cd $R
python - <<PY
import csv, glob, json, collections
def per_dispatch(counter):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % counter)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter and "conv_igemm" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]
fetch, write = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
# bench_conv: 8 shapes x (2 warm-up + 2 timed) launches, in order
shapes = ["pers L0 320->320", "pano L0 320->320 (W+4)", "pers L1 640->640", "pers L2 1280->1280", "pers L3 1280->1280",
          "pers up L0 960->320", "pers up L1 1920->640", "pano L0 wrap s1"]
out = {}
n = min(len(fetch), len(write)) // len(shapes)
for i, s in enumerate(shapes):
    fk = sum(fetch[i * n:(i + 1) * n]) / n
    wk = sum(write[i * n:(i + 1) * n]) / n
    # gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B -> x2 (MI355X_MICROARCH.md, HBM)
    out[s] = {"fetch_kb_raw": fk, "write_kb": wk, "traffic_bytes": (2.0 * fk + wk) * 1024.0}
json.dump(out, open("$OUT/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
