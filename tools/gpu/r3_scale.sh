set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_scale.log
: > $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_GUARD_SCALE=1,0.5,0.25,0 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
unset SVS_GUARD_SCALE
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
cat $L
