#!/bin/bash
# Round 4, step 2: full GPU test tier, A/B against the round-2 library (fast mode: what that library's flags mean), pooled
# vs wave-private replay on natural content, 16-byte loads in the one-row extract kernel.
set -u
mkdir -p gpurun_out/r4s2
export TMPDIR=/tmp
E=gpurun_out/r4s2
V=secure-video-steganography-using-ecc-and-dct_amd/lib
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; return $rc; }
step 1100 pytest_gpu.log python -m pytest tests -x -q -m gpu --durations=8 || { tail -40 $E/pytest_gpu.log; exit 1; }
tail -14 $E/pytest_gpu.log
: > $E/ab_vs_r02.txt
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== ab $cfg (flags 0): this build vs the round-2 library" >> $E/ab_vs_r02.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -E "frames|embed med|read_nt|copy16_nt " >> $E/ab_vs_r02.txt
done
cat $E/ab_vs_r02.txt
echo "== one-row exact extract: 8-byte vs 16-byte row loads" > $E/ab_extract_bpl.txt
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 --env-sweep SVS_EXTRACT_EXACT_BPL=1,2 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "frames|embed med" >> $E/ab_extract_bpl.txt
timeout -k 10 300 python tools/ab_bench.py --frames 300 --h 1080 --w 1920 --n-ac 3 --rounds 9 --env-sweep SVS_EXTRACT_EXACT_BPL=1,2 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "frames|embed med" >> $E/ab_extract_bpl.txt
cat $E/ab_extract_bpl.txt
step 300 probe_base.txt python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,dark
SVSDCT_LIB=$V/variants/libsvsdct_pool.so timeout -k 10 300 python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,dark > $E/probe_pool.txt 2>&1
grep -v amdgpu $E/probe_base.txt; echo "-- pooled:"; grep -v amdgpu $E/probe_pool.txt
