#!/bin/bash
# GPU clocks / power while a command runs: tools/clock_watch.sh <out file> <command...>   (GPU box)
# polls rocm-smi every 0.25 s: is the shader clock under the bench's sustained load the clock of a short burst?
OUT=$1; shift
"$@" > $OUT.cmd.log 2>&1 &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> $OUT
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use|Temperature \(Sensor (edge|junction|hotspot)" >> $OUT
  sleep 0.25
done
wait $PID
