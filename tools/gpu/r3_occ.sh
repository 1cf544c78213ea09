set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_occ.log
: > $L
echo "== n=3: base (U1 min waves 5, scratch) vs u1w4 (100 VGPRs, no scratch) vs r02" >> $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_u1w4.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
echo "== n=10: base (U2 min waves 5) vs u2w6 (80 VGPRs, scratch) vs r02" >> $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 10 --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_u2w6.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
echo "== n=3 BPL=1 occupancy sweep" >> $L
SVS_EMBED_BPL=1 timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,5,6,7 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
echo "== n=3 chunk sweep" >> $L
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --chunks 4294967295,16,32,64 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
cat $L
