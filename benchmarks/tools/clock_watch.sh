#!/bin/bash
# Sample the shader clock, memory clock and socket power while the bench loop runs (GPU box):
#   benchmarks/tools/clock_watch.sh [steps]
cd $GRAFT_REPO_ROOT
steps=${1:-3000}
python bench.py --no-cpu-baseline --no-single --steps $steps --warmup 5 > /tmp/bench_watch.log 2>&1 &
pid=$!
sleep 20   # import + input generation + upload
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ' '
  echo
  sleep 1
done
wait $pid
tail -1 /tmp/bench_watch.log | cut -c1-200
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '
