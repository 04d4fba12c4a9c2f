cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -c 600 gpurun_out/r02/bench_default.err
python bench.py --engine python --no-cpu-baseline --no-single-stream --no-deferred > gpurun_out/r02/bench_python_engine.json 2>> gpurun_out/r02/bench_default.err
