#!/bin/bash
# rocprofv3 kernel-trace statistics of the benchmark command; keeps only the small summaries.
#   tools/profile_bench.sh <out_dir_under_gpurun_out> [bench args...]
set -u
R=$PWD
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py "$@" > $OUT/bench_stdout.log 2>&1
cd $R
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
find $OUT -type f | head -20
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
tail -2 $OUT/bench_stdout.log | cut -c1-600
