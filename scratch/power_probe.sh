#!/bin/bash
# power / clocks while a loop of evaluations runs: bash scratch/power_probe.sh <workload>
R=$GRAFT_REPO_ROOT
cd $R
python scratch/wall_loop.py $1 ${2:-1500} > /tmp/wl.out 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | head -8
  echo ---
  sleep 0.7
done
wait $PID
tail -1 /tmp/wl.out
