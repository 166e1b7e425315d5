#!/bin/bash
# visit 4r: clock and duration of every conv launch inside a steady-state fp32 forward (tools/sclk_per_layer.py)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/sclk_per_layer.py --forwards 100 > gpurun_out/4r_sclk_per_layer.txt 2>&1 || { tail -20 gpurun_out/4r_sclk_per_layer.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/4r_sclk_per_layer.txt
