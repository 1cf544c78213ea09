set -u
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_u1w5.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids"
