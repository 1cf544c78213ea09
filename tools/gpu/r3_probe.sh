#!/bin/bash
set -u
E=gpurun_out/ev; mkdir -p $E
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; }
step 600 guarded_probe_n3.txt python tools/guarded_probe.py --frames 200 --json $E/guarded_probe_n3.json
step 300 guarded_probe_n1.txt python tools/guarded_probe.py --frames 200 --n-ac 1 --classes noise,natural,flat128,letterbox25
step 300 guarded_probe_n7.txt python tools/guarded_probe.py --frames 200 --n-ac 7 --classes noise,natural,flat128,letterbox25
step 400 guarded_probe_n10.txt python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,flat128,letterbox25,rows,checker8,dark,bright
step 300 guarded_probe_n15.txt python tools/guarded_probe.py --frames 200 --n-ac 15 --delta 20 --classes noise,natural
grep -hv amdgpu.ids $E/guarded_probe_n*.txt
