set -u
mkdir -p gpurun_out
V=secure-video-steganography-using-ecc-and-dct_amd/lib
L=gpurun_out/ab_u2b.log
: > $L
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 600 --n-ac 10 --delta 20"; do
  echo "== $cfg: base(u2w5) u2w6 r02" >> $L
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_u2w6.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids" >> $L
done
cat $L
