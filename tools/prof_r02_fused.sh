cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x > gpurun_out/r02/gpu_tests_fused.log 2>&1; tail -3 gpurun_out/r02/gpu_tests_fused.log
python bench.py --defer 0 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r02/bench_strict_fused.json 2> gpurun_out/r02/bench_strict_fused.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-single-stream > gpurun_out/r02/bench_defer_fused.json 2>> gpurun_out/r02/bench_strict_fused.err
SC_DEC_FUSED=0 python bench.py --defer 0 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r02/bench_strict_unfused.json 2>> gpurun_out/r02/bench_strict_fused.err
rm -rf /tmp/p1; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p1 -- python3 bench.py --defer 0 --steps 6 --warmup 3 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/prof_strict_fused.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r02/strict_fused_kernel_stats.csv > /dev/null
for w in -3 -40 -80 -120; do python tools/rocpd_timeline.py $DB $w 120 > gpurun_out/r02/strict_fused_timeline_$w.txt 2>&1; done
