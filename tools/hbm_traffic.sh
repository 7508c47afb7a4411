#!/bin/bash
# HBM traffic of the conv / GEMM kernels at the cfg2 shapes (tools/hbm_traffic.py; rocprofv3 PMC, two separate passes as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Do NOT point this at bench.py: under
# --pmc the model build alone (tens of thousands of tiny init kernels) takes tens of minutes.
#   tools/hbm_traffic.sh <out_dir_under_gpurun_out> [halo]
set -u
R=$PWD
OUT=$R/gpurun_out/$1
MODE=${2:-}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- \
    python $R/tools/hbm_traffic.py run $OUT/manifest.json $MODE > $OUT/$C.log 2>&1
  find $OUT/$C -name "*kernel_trace.csv" -delete
done
cd $R
python tools/hbm_traffic.py parse $OUT
