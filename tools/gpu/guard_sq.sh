set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
SVS_DCT_MODE=guarded TAG=guard BENCH_ARGS="--frames 600" bash tools/gpu_pmc_sq.sh > gpurun_out/guard_sq_run.log 2>&1
python tools/sq_summary.py guard > gpurun_out/guard_sq_summary.txt 2>&1
TAG=fast BENCH_ARGS="--frames 600" bash tools/gpu_pmc_sq.sh >> gpurun_out/guard_sq_run.log 2>&1
python tools/sq_summary.py fast >> gpurun_out/guard_sq_summary.txt 2>&1
cat gpurun_out/guard_sq_summary.txt
