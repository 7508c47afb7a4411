#!/bin/bash
# CPU baseline of BASELINE.md section 4 (host cores of the GPU box, no GPU work) + the other configurations as single-GPU steps
O=gpurun_out
for w in cfg1 cfg4 cfg5; do
  dt=bf16; [ $w = cfg5 ] && dt=fp16
  timeout 600 python bench.py --workload $w --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $O/b5_$w.json 2> $O/b5_$w.err
  tail -c 300 $O/b5_$w.json; echo
done
timeout 1500 python tools/cpu_baseline.py > $O/b5_cpu_baseline.jsonl 2> $O/b5_cpu_baseline.err
cut -c1-700 $O/b5_cpu_baseline.jsonl; tail -3 $O/b5_cpu_baseline.err
