export TMPDIR=/tmp
O=gpurun_out/c5sweep; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --workload c5-shard --steps 100 --warmup 10 --no-cpu-baseline --no-c5-entry > $O/bench.json 2> $O/err.txt
python3 tools/sweep_timeline.py $O/trace > $O/timeline.txt 2>&1
rm -rf $O/trace
cat $O/timeline.txt
