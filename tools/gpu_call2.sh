#!/bin/bash
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -k "whole_denoising_step or masks_at_the_cfg5 or test_dist_gpu" > $O/b2_tests.log 2>&1
echo "tests rc=$?" >> $O/b2_tests.log
tail -6 $O/b2_tests.log; grep -E "PARITY|sharded vs" $O/b2_tests.log | cut -c1-900
IM360_KERNELS_LIB=$PWD/imagine360_amd/libim360_kernels_ablate.so timeout 600 python tools/ab_stag.py --iters 10 > $O/b2_a3cm.log 2>&1; cat $O/b2_a3cm.log | cut -c1-250
timeout 300 python tools/ab_groups.py --iters 10 > $O/b2_groups.log 2>&1; cat $O/b2_groups.log | cut -c1-300
timeout 300 python tools/dual_stream_race.py --tuned --runs 15 > $O/b2_race.log 2>&1; tail -6 $O/b2_race.log
