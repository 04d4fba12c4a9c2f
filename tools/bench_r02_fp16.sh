cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_native.py -m gpu -q -x -k "fp16" 2>&1 | tail -15
python bench.py --kv-dtype float16 --no-cpu-baseline --no-deferred > gpurun_out/r02/bench_kv_fp16.json 2> gpurun_out/r02/bench_kv_fp16.err
python bench.py --no-cpu-baseline --no-deferred > gpurun_out/r02/bench_kv_fp32.json 2>> gpurun_out/r02/bench_kv_fp16.err
