#!/bin/bash
# Where do the ~5.6 us dispatch gaps of the round-3 timelines come from?  One step's kernel rows (every column) under
# rocprofv3 for: the default build, bf16x3 off; plus the unprofiled step time of each.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/gap
rm -rf $O; mkdir -p $O
cd $R
run() {   # tag, extra bench flags (environment set by the caller)
  tag=$1; shift
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-c5-entry "$@" 2> $O/bench_$tag.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', 'us/step', 1e3 * d['ms_per_step'])" >> $O/summary.txt
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c5-entry "$@" > /dev/null 2> $O/rocprof_$tag.err
  python3 tools/gap_probe.py $O/trace_$tag > $O/probe_$tag.txt 2>&1
  rm -rf $O/trace_$tag
}
run default
SMX_TUNING=bf16x3=0 run no_bf16x3
SMX_TUNING=no_adam_early run no_adam_early
cat $O/summary.txt
