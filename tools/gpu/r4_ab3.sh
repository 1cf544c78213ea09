#!/bin/bash
set -u
mkdir -p gpurun_out/r4ab3
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -x -q -m gpu -k "guarded or exact_mode or fast_mode or golden or pitched or random_geometries or stateless or knobs" 2>&1 | tail -3 || exit 1
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 600 --n-ac 12 --delta 20"; do
  echo "== $cfg guarded: base (rows parked in LDS, phase 1 in place: 68 VGPRs, 5 waves) | noinplace (103 VGPRs, 4 waves)"
  timeout -k 10 400 python tools/ab_bench.py $cfg --mode guarded --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_noinplace.so 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*| / | /'
done > gpurun_out/r4ab3/ab.txt 2>&1
cat gpurun_out/r4ab3/ab.txt
timeout -k 10 300 python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,dark 2>&1 | grep -v amdgpu
