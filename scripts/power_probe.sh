#!/bin/bash
# Socket power and clocks while a workload runs: rocm-smi sampled every 0.5 s beside it.  Usage: power_probe.sh <label> <command...>
label=$1; shift
"$@" > /tmp/pp_$label.log 2>&1 &
pid=$!
sleep 4
for i in $(seq 1 12); do
  if ! kill -0 $pid 2>/dev/null; then break; fi
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $pid
tail -1 /tmp/pp_$label.log
