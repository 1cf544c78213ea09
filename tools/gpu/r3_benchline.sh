#!/bin/bash
set -u
E=gpurun_out/ev; mkdir -p $E; export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $E/bench.json 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_stats2 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 > $E/rocprof_stats.log 2>&1 || exit 1
cp $E/prof_stats2/*/*_kernel_stats.csv $E/kernel_stats.csv
grep '^{' $E/bench.json | tail -1 | cut -c1-1500
