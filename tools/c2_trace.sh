#!/bin/bash
# development: rocprofv3 kernel trace of the C2 workload -- per-kernel averages and the timeline of one step (tools/prof_summary.py).  usage: c2_trace.sh [workload]
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
W=${1:-8kly}
O=gpurun_out/c2trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload $W --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry > $O/bench.json 2> $O/err.txt
python3 tools/prof_summary.py $O/trace > $O/summary.txt 2>&1
rm -rf $O/trace
grep -A14 "one step" $O/summary.txt
