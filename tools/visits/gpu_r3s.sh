#!/bin/bash
# visit s: shader clock inside every fp32 conv launch
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/sclk_per_layer.py > gpurun_out/s_sclk.log 2>&1 || { tail -20 gpurun_out/s_sclk.log; exit 1; }
cat gpurun_out/s_sclk.log
