python -m pytest tests -x -q -m gpu -k "marginal or score or llk or predict" > gpurun_out/s3_t.log 2>&1; tail -3 gpurun_out/s3_t.log
bash tools/_run1.sh
