set -u
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_patch.log
: > $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --classes noise,natural,flat128,letterbox25,checker8,bright 2>&1 | grep -v amdgpu | tee -a $L
timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 1 --classes noise,flat128 2>&1 | grep -v amdgpu | tee -a $L
echo "== ab: base (patch kernel) / SVS_EMBED_PATCH=0 (register-holding kernel)" | tee -a $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 --env-sweep SVS_EMBED_PATCH=1,0 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" | tee -a $L
unset SVS_EMBED_PATCH
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" | tee -a $L
