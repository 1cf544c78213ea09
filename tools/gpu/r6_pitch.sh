#!/bin/bash
set -u
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
timeout -k 10 300 python tools/pitch_probe.py --pairs 4 --cap 4 > gpurun_out/r6c/pitch_cap4.txt 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r6c/pitch_cap4.txt
timeout -k 10 300 python tools/pitch_probe.py --pairs 4 --cap 0 > gpurun_out/r6c/pitch_cap0.txt 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r6c/pitch_cap0.txt
