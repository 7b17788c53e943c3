#!/bin/bash
# kernel trace of marginal_llk calls (tools/score_pmc.py, DRAWS=100 CALLS via the script's two calls) and the timeline of one call
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc_tr
CALLS=6 DRAWS=100 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc_tr -- python3 $GRAFT_REPO_ROOT/tools/score_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/score_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/score_timeline.py /tmp/sc_tr >> $GRAFT_REPO_ROOT/gpurun_out/score_trace.log 2>&1
tail -30 $GRAFT_REPO_ROOT/gpurun_out/score_trace.log
