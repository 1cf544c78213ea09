set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
: > gpurun_out/ab_u2.log
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 10 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_u2w6.so $V/variants/libsvsdct_u2w4.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> gpurun_out/ab_u2.log
cat gpurun_out/ab_u2.log
TAG=n10new BENCH_ARGS="--frames 600 --n-ac 10" bash tools/gpu_pmc_sq.sh > gpurun_out/sq_n10_run.log 2>&1
python tools/sq_summary.py n10new > gpurun_out/sq_n10new_summary.txt 2>&1
SVSDCT_LIB=$V/variants/libsvsdct_r02.so TAG=n10r02 BENCH_ARGS="--frames 600 --n-ac 10" bash tools/gpu_pmc_sq.sh >> gpurun_out/sq_n10_run.log 2>&1
python tools/sq_summary.py n10r02 > gpurun_out/sq_n10r02_summary.txt 2>&1
grep -A3 "^embed_kernel" gpurun_out/sq_n10new_summary.txt gpurun_out/sq_n10r02_summary.txt
