set -e
mkdir -p gpurun_out
L=gpurun_out/guard_scope.log
: > $L
V=secure-video-steganography-using-ecc-and-dct_amd/lib/variants
for lib in "" $V/libsvsdct_gscope0.so; do
  echo "== lib=${lib:-default(scope1)}" >> $L
  SVSDCT_LIB=${lib:-secure-video-steganography-using-ecc-and-dct_amd/lib/libsvsdct.so} timeout -k 10 300 python tools/guarded_probe.py --frames 200 --classes noise,natural,flat128,letterbox25 >> $L 2>&1
done
echo "== scope1 n=1 and n=7" >> $L
timeout -k 10 200 python tools/guarded_probe.py --frames 200 --n-ac 1 --classes noise,flat128 >> $L 2>&1
timeout -k 10 200 python tools/guarded_probe.py --frames 200 --n-ac 7 --classes noise,flat128 >> $L 2>&1
echo "== scope1 600 frames" >> $L
timeout -k 10 300 python tools/guarded_probe.py --frames 600 --classes noise >> $L 2>&1
grep -v amdgpu.ids $L
