# development: rocprofv3 kernel stats of `bench.py --workload c5-shard` + one step's timeline (and the sweep's, queue by queue)
export TMPDIR=/tmp
O=gpurun_out/c5prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload c5-shard --steps 100 --warmup 10 --no-cpu-baseline --no-c5-entry > $O/bench.json 2> $O/err.txt
python3 tools/prof_summary.py $O/trace > $O/summary.txt 2>&1
python3 tools/sweep_timeline.py $O/trace > $O/sweep_timeline.txt 2>&1
rm -rf $O/trace
cat $O/summary.txt | tail -40
tail -40 $O/sweep_timeline.txt
