set -u
V=secure-video-steganography-using-ecc-and-dct_amd/lib
echo "== patch kernel, guard scale sweep"
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_GUARD_SCALE=1,0.25,0 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids"
unset SVS_GUARD_SCALE
echo "== register-holding kernel, guard scale sweep"
SVS_EMBED_PATCH=0 timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_GUARD_SCALE=1,0.25,0 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids"
unset SVS_GUARD_SCALE
echo "== patch kernel occupancy sweep"
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,5,4,3 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids"
echo "== r02"
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" | head -3
