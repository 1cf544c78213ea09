set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/guarded3.log
: > $L
timeout -k 10 300 python tools/guarded_probe.py --frames 200 --n-ac 3 --classes noise,natural,flat128,letterbox25 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 10 --classes noise,natural,flat128,letterbox25,dark 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" | tee -a $L
timeout -k 10 300 python tools/tie_fallback_rate.py 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
