set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_occ2.log
: > $L
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 200 --n-ac 63"; do
  echo "== ab $cfg: base vs u2w6 vs r02" >> $L
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_u1w4_u2w6.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
done
cat $L
timeout -k 10 600 python tools/guarded_probe.py --frames 200 --json gpurun_out/guarded_probe_n3.json > gpurun_out/guarded_probe_n3.log 2>&1
grep -v amdgpu.ids gpurun_out/guarded_probe_n3.log
