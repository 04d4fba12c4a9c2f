cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python bench.py --defer 0 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r02/bench_strict_base.json 2> gpurun_out/r02/bench_strict_base.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-single-stream > gpurun_out/r02/bench_defer_base.json 2>> gpurun_out/r02/bench_strict_base.err
rm -rf /tmp/p1; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p1 -- python3 bench.py --defer 0 --steps 6 --warmup 3 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/prof_strict.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r02/strict_kernel_stats.csv > /dev/null
for w in -3 -40 -80 -120; do python tools/rocpd_timeline.py $DB $w 120 > gpurun_out/r02/strict_timeline_$w.txt 2>&1; done
rm -rf /tmp/p2; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p2 -- python3 bench.py --streams 1 --defer 0 --steps 10 --warmup 4 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/prof_s1.log 2>&1
DB=$(find /tmp/p2 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r02/s1_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $DB -3 120 > gpurun_out/r02/s1_timeline.txt 2>&1
