set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=n63d8 BENCH_ARGS="--frames 200 --n-ac 63 --delta 8" bash tools/gpu_pmc_sq.sh > gpurun_out/sq_n63_run.log 2>&1
python tools/sq_summary.py n63d8 > gpurun_out/sq_n63d8_summary.txt 2>&1
TAG=n63d20 BENCH_ARGS="--frames 200 --n-ac 63 --delta 20" bash tools/gpu_pmc_sq.sh >> gpurun_out/sq_n63_run.log 2>&1
python tools/sq_summary.py n63d20 > gpurun_out/sq_n63d20_summary.txt 2>&1
TAG=n20d8 BENCH_ARGS="--frames 200 --n-ac 20 --delta 8" bash tools/gpu_pmc_sq.sh >> gpurun_out/sq_n63_run.log 2>&1
python tools/sq_summary.py n20d8 > gpurun_out/sq_n20d8_summary.txt 2>&1
cat gpurun_out/sq_n63d8_summary.txt gpurun_out/sq_n63d20_summary.txt gpurun_out/sq_n20d8_summary.txt | grep -v "instruction"
