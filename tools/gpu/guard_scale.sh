set -e
mkdir -p gpurun_out
for sc in 0 0.25 1 4 16; do
  echo "== SVS_GUARD_SCALE=$sc" >> gpurun_out/guard_scale.log
  SVS_GUARD_SCALE=$sc timeout -k 10 200 python tools/guarded_probe.py --frames 200 --classes noise,natural,flat128 >> gpurun_out/guard_scale.log 2>&1
done
for k in 3 4 5; do
  echo "== SVS_EMBED_WG_PER_CU=$k" >> gpurun_out/guard_scale.log
  SVS_EMBED_WG_PER_CU=$k timeout -k 10 200 python tools/guarded_probe.py --frames 200 --classes noise >> gpurun_out/guard_scale.log 2>&1
done
for c in 0 16 32 64; do
  echo "== SVS_EMBED_XCD_CHUNK=$c" >> gpurun_out/guard_scale.log
  SVS_EMBED_XCD_CHUNK=$c timeout -k 10 200 python tools/guarded_probe.py --frames 200 --classes noise >> gpurun_out/guard_scale.log 2>&1
done
echo "== 600 frames" >> gpurun_out/guard_scale.log
timeout -k 10 300 python tools/guarded_probe.py --frames 600 --classes noise >> gpurun_out/guard_scale.log 2>&1
grep -v amdgpu.ids gpurun_out/guard_scale.log
