#!/bin/bash
# Regenerate imagine360_amd/tuning/tunableop_gfx950_cfg2_bf16.csv on an MI355X (takes ~5 GPU-minutes):
#   tools/tune_gemms.sh     (run through gpurun; copies the table to gpurun_out/)
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --cpu-baseline none --no-graph --no-tuned-gemms
