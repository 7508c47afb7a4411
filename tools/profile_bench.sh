#!/bin/bash
# rocprofv3 kernel-trace statistics of the benchmark command, for the code that SHIPS: two passes
#   1) `bench.py --dual-stream 0`: every kernel alone on the chip -- the summary whose average duration for the dominant kernel must
#      agree with the `roofline` block of the bench line (bench.py takes its per-class durations from a one-stream pass as well);
#      the per-shape table of one eager step (tools/trace_by_shape.py) comes from this pass's trace;
#   2) the default command (panorama branch on a side stream): the timed launch mode; kernel durations there include the time a
#      kernel shares the chip with the other branch.
# Kernel names stay MANGLED (-M) so that tests/test_host_logic.py can check every im360 symbol of the newest summary against the
# kernels inside the shipped library.  Only the small summaries are kept.
#   tools/profile_bench.sh <out_dir_under_gpurun_out> <commit hash> [bench args...]
# Afterwards (authoring container): copy <out>/<round>_<hash>_* into profiles/.
set -u
R=$PWD
OUT=$R/gpurun_out/$1; shift
HASH=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
LIBSHA=$(sha256sum $R/imagine360_amd/libim360_kernels.so | cut -c1-16)
for mode in 1stream 2stream; do
    D=$OUT/$mode
    mkdir -p $D
    extra=""
    [ $mode = 1stream ] && extra="--dual-stream 0"
    cd /tmp
    rocprofv3 --kernel-trace --stats -M --output-format csv -d $D -o bench -- python $R/bench.py --no-cpu-baseline $extra "$@" > $D/bench_stdout.log 2>&1
    cd $R
    stats=$(find $D -name "*kernel_stats.csv" | head -1)
    trace=$(find $D -name "*kernel_trace.csv" | head -1)
    tag=${ROUND:-r06}_${HASH}_bench_cfg2$([ $mode = 1stream ] && echo _1stream)
    if [ -n "$stats" ]; then
        { echo "# rocprofv3 --kernel-trace --stats -M -- python bench.py --no-cpu-baseline $extra $*   (commit $HASH, libim360_kernels.so sha256 $LIBSHA)"; cat "$stats"; } > $OUT/${tag}_kernel_stats.csv
    fi
    grep '^{"metric"' $D/bench_stdout.log | tail -1 > $OUT/${tag}_profiled.json
    if [ $mode = 1stream ] && [ -n "$trace" ]; then
        python $R/tools/trace_by_shape.py "$trace" --steps 1 --skip-steps 1 > $OUT/${ROUND:-r06}_${HASH}_step_by_shape.txt 2>> $D/bench_stdout.log
    fi
    find $D -name "*kernel_trace.csv" -delete
    find $D -name "*.db" -delete
    tail -3 $D/bench_stdout.log | cut -c1-400
done
ls -la $OUT | head -20
head -12 $OUT/${ROUND:-r06}_${HASH}_bench_cfg2_1stream_kernel_stats.csv | cut -c1-200
