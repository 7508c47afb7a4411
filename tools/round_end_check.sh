#!/bin/bash
# What the builder runs on a GPU box before handing a round over (the driver repeats the first three itself):
#   smoke(), a subset of the GPU tests that touches every kernel family, the driver's bench command, the rocprofv3 summaries of the
#   commit (tools/profile_bench.sh) and -- optionally -- the CPU baseline of the benchmarked workload measured directly.
#   tools/round_end_check.sh <commit hash> [cpu-threads for the direct cfg2 baseline, 0 = skip]
HASH=$1
CPU=${2:-0}
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/end_smoke.log 2>&1; tail -2 $O/end_smoke.log
timeout 900 python -m pytest tests -m gpu -q -k "library_loads or test_linear or geglu or gemm_pipeline or test_conv_variants or test_group_norm or test_attention_pipelined or temporal_attention or dual_stream_forward" > $O/end_tests.log 2>&1
echo "tests rc=$?" >> $O/end_tests.log; tail -3 $O/end_tests.log
timeout 900 bash tools/profile_bench.sh end_prof $HASH --steps 10 --warmup 3 > $O/end_prof.log 2>&1; tail -4 $O/end_prof.log | cut -c1-200
timeout 600 python bench.py --steps 20 --warmup 5 > $O/end_bench.json 2> $O/end_bench.err; cut -c1-330 $O/end_bench.json
if [ "$CPU" != "0" ]; then
  timeout 1200 python tools/cpu_baseline.py --cfg2-threads $CPU > $O/end_cpu_cfg2.json 2> $O/end_cpu_cfg2.err; cut -c1-900 $O/end_cpu_cfg2.json
fi
