set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_wgscope.log
: > $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_wgscope.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 10 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_wgscope.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
cat $L
SVSDCT_LIB=$V/variants/libsvsdct_wgscope.so timeout -k 10 300 python tools/guarded_probe.py --frames 100 --classes noise,flat128,letterbox25,natural 2>&1 | grep -v amdgpu
SVSDCT_LIB=$V/variants/libsvsdct_wgscope.so timeout -k 10 300 python tools/guarded_probe.py --frames 100 --n-ac 1 --classes noise,flat128 2>&1 | grep -v amdgpu
