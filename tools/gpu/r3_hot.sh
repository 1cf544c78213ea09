set -u
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 15 $V/libsvsdct.so $V/variants/libsvsdct_hot.so $V/variants/libsvsdct_hot6.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "copy \|amdgpu.ids"
done
