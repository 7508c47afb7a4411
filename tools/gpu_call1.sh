#!/bin/bash
# round-5 GPU call 1: the new / changed tests, the two-stream race diagnosis, rocprofv3 summaries of the shipped library, one bench line
HASH=$1
O=gpurun_out
rm -f $O/parity_observed.json
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "whole_denoising_step or masks_at_the_cfg5 or cfg2_step_vs_reference or test_dist_gpu or dual_stream_forward or test_abi" > $O/b1_tests.log 2>&1
echo "tests rc=$?" >> $O/b1_tests.log
tail -5 $O/b1_tests.log
grep PARITY $O/b1_tests.log | cut -c1-700
timeout 600 python tools/dual_stream_race.py --tuned --runs 5 > $O/b1_race.log 2>&1; tail -12 $O/b1_race.log
timeout 900 bash tools/profile_bench.sh b1_prof $HASH --steps 10 --warmup 3 > $O/b1_prof.log 2>&1; tail -15 $O/b1_prof.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > $O/b1_bench.json 2> $O/b1_bench.err; tail -c 1500 $O/b1_bench.json
