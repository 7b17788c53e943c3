export TMPDIR=/tmp
O=gpurun_out/sct; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-entry > $O/bench.json 2> $O/err.txt
python3 tools/score_timeline.py $O/trace
rm -rf $O/trace
