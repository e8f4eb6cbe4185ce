#!/bin/bash
# GPU box: samples rocm-smi (socket power, sclk) every 0.25 s while the c3 step runs back to back (bench.py with many steps),
# then once more idle.  Output: gpurun_out/power_probe.txt.   usage: bash tools/micro/power_probe.sh
R=$PWD; O=$R/gpurun_out; mkdir -p $O
: > $O/power_probe.txt
python3 $R/bench.py --steps 2500 --warmup 20 --no-cpu-baseline > $O/power_probe_bench.json 2> $O/power_probe_bench.err &
BP=$!
sleep 6      # import + build of the layers + warm-up
for i in $(seq 1 24); do
  echo "--- sample $i (step running)" >> $O/power_probe.txt
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|temperature (sensor junction)\|junction" >> $O/power_probe.txt
  sleep 0.25
done
wait $BP
sleep 2
echo "--- idle" >> $O/power_probe.txt
rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" >> $O/power_probe.txt
rocm-smi --showmaxpower 2>&1 | grep -i "max" >> $O/power_probe.txt
tail -c 300 $O/power_probe_bench.json
