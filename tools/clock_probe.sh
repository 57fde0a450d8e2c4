#!/bin/bash
# Sample the GPU's clock and power (rocm-smi) while a command runs: bash tools/clock_probe.sh <out.txt> <command...>
OUT=$1; shift
"$@" > $OUT.cmd.log 2>&1 &
PID=$!
sleep ${PROBE_DELAY:-25}
for i in $(seq ${PROBE_N:-12}); do
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' ' | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.5
done
wait $PID
tail -1 $OUT.cmd.log | cut -c1-200 >> $OUT
