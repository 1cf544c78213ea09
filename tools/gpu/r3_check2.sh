set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
: > gpurun_out/ab_r02_vs_r03.log
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 200 --n-ac 63" "--frames 200 --n-ac 20"; do
  echo "== ab $cfg" >> gpurun_out/ab_r02_vs_r03.log
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> gpurun_out/ab_r02_vs_r03.log
done
echo "== occupancy sweep n=3 (new lib)" >> gpurun_out/ab_r02_vs_r03.log
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,4,5,6,7,8 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" >> gpurun_out/ab_r02_vs_r03.log
echo "== BPL sweep n=3 (new lib)" >> gpurun_out/ab_r02_vs_r03.log
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_BPL=2,1 $V/libsvsdct.so 2>&1 | grep -v "copy \|amdgpu.ids" >> gpurun_out/ab_r02_vs_r03.log
cat gpurun_out/ab_r02_vs_r03.log
