set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 200 --n-ac 63" "--frames 200 --n-ac 20"; do
  echo "== ab $cfg" >> gpurun_out/ab_r02_vs_r03.log
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> gpurun_out/ab_r02_vs_r03.log
done
cat gpurun_out/ab_r02_vs_r03.log
