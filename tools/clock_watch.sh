#!/bin/bash
# Sample shader clock / power / temperature with rocm-smi while a command runs (is the chip holding 2.4 GHz under the
# sustained fp32-MFMA load the roofline peak assumes?).  Usage: tools/clock_watch.sh OUT.log -- cmd args...
out=$1; shift; shift
( while true; do
    date +%s.%N
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" 
    sleep 0.3
  done ) > "$out" 2>&1 &
watcher=$!
"$@"
rc=$?
kill $watcher 2>/dev/null
wait $watcher 2>/dev/null
exit $rc
