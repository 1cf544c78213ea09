set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/tie_fallback.log
: > $L
echo "== new library" >> $L
timeout -k 10 300 python tools/tie_fallback_rate.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== round-2 library" >> $L
SVSDCT_LIB=$V/variants/libsvsdct_r02.so SVS_SKIP_ABI_CHECK=1 timeout -k 10 300 python tools/tie_fallback_rate.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "natural_like or structured or golden or random_geometries or extreme" > gpurun_out/pytest_part.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_part.log
