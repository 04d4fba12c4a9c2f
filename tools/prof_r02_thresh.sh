cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "ragged or distinct or full_size or xl" > gpurun_out/r02/quick_tests2.log 2>&1; tail -3 gpurun_out/r02/quick_tests2.log
for t in 0 80 160 320 640 1280; do
SC_DEC_FUSED_MAX_ROWS=$t python bench.py --defer 0 --steps 10 --warmup 4 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/bench_strict_t$t.json 2>> gpurun_out/r02/bench_t.err
SC_DEC_FUSED_MAX_ROWS=$t python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/bench_defer_t$t.json 2>> gpurun_out/r02/bench_t.err
done
