#!/bin/bash
# Power draw and clocks while the timed passes run (rocm-smi samples every 0.5 s beside a long bench run), then idle.
python bench.py --steps 1500 --warmup 5 --no-cpu --no-learn --no-f64 --no-host-learn > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
sleep 4
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ';' ; echo
  sleep 0.7
done
wait $pid
tail -c 400 gpurun_out/power_bench.json; echo
sleep 2
echo idle:; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" ; rocm-smi --showpowercap 2>/dev/null | grep -i cap | head -3
