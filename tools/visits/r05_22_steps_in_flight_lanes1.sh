#!/bin/bash
# round 5, visit 22: does running DIFFERENT phases of the network side by side return anything?  D whole-batch nets with ONE lane each (one stream = one hardware
# queue per net, no queue shared), started a fraction of a step apart, against the lock-step sub-batch lanes of one net (same number of kernels in flight)
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_22_steps_in_flight_lanes1.txt
: > $O
for spec in "bf16 128 --graph" "f32 64"; do
  set -- $spec
  for lanes in 1 2; do
    timeout -k 10 400 python tools/gate_steps_in_flight.py --dtype $1 --batch $2 $3 --lanes $lanes --depths 1,2,3 --offsets 0,0.25,0.5 --steps 48 --rounds 1 >> $O 2> gpurun_out/r05_22.err || { tail -20 gpurun_out/r05_22.err; exit 1; }
  done
  timeout -k 10 400 python tools/gate_steps_in_flight.py --dtype $1 --batch $2 $3 --lanes 3 --depths 1 --steps 48 --rounds 1 >> $O 2> gpurun_out/r05_22.err || { tail -20 gpurun_out/r05_22.err; exit 1; }
done
cat $O
