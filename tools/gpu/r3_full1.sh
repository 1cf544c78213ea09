set -u
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench_n1.log 2>&1; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_n1.log
timeout -k 10 400 python tools/many_coefficients.py > gpurun_out/many_coefficients.txt 2>&1; echo "many rc=$?"; grep -v amdgpu.ids gpurun_out/many_coefficients.txt
