set -u
mkdir -p gpurun_out
L=gpurun_out/guarded2.log
: > $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 10 --classes noise,natural,flat128,letterbox25,rows,checker8,dark,bright 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 15 --delta 20 --classes noise,natural 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 8 --delta 4 --classes noise,natural,flat128 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/guarded_probe.py --frames 600 --n-ac 10 --classes noise 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "guarded or exact_mode or stateless" > gpurun_out/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_part.log
