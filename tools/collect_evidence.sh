#!/bin/bash
# Copy what tools/gpu/r6_evidence.sh (rounds 4, 5: r4_evidence.sh, r5_evidence.sh - in git history) left under gpurun_out/ev into profiles/ (run in the build container, where .git is):
#   tools/collect_evidence.sh r04_c      -> profiles/r04_c_bench.json, ..., and the un-suffixed counter / probe / A/B files
# Every kernel-stats file is headed by the commit and by the kernel-source hash bench.py computes.
set -eu
TAG=${1:?tag, e.g. r03_c}; ROUND=${TAG%%_*}
cd "$(dirname "$0")/.."
E=gpurun_out/ev; REV=$(git rev-parse --short HEAD)
grep '^{' $E/bench.json | tail -1 > profiles/${TAG}_bench.json
SHA=$(python -c "import bench; print(bench.kernel_source_sha())")
stats() { { echo "# commit $REV, kernel sources sha256 $SHA"; echo "# $2"; cat "$1"; } > "$3"; }
stats $E/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0   (600 x 3840x2160, n = 3, delta = 8, guarded = the default mode)" profiles/${TAG}_kernel_stats.csv
[ -f $E/kernel_stats_g10.csv ] && stats $E/kernel_stats_g10.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10   (600 x 3840x2160, n = 10: the two-row rigorous kernel embed_kernel<2, QM, 1, 10> and extract_kernel<2, QM, 1, 10>)" profiles/${TAG}_kernel_stats_n10.csv
[ -f $E/kernel_stats_g10_1080.csv ] && stats $E/kernel_stats_g10_1080.csv "same, --frames 300 --height 1080 --width 1920 (BASELINE configs[1])" profiles/${TAG}_kernel_stats_n10_1080p.csv
[ -f $E/kernel_stats_g10_1080x2400.csv ] && stats $E/kernel_stats_g10_1080x2400.csv "same, --frames 2400 --height 1080 --width 1920: the pixel count of 600 x 4K in 1080p frames (is the 1080p rate a matter of the frame geometry or of the launch size?)" profiles/${TAG}_kernel_stats_n10_1080p_x2400.csv
[ -f $E/kernel_stats_colour_n10.csv ] && stats $E/kernel_stats_colour_n10.csv "rocprofv3 --kernel-trace --stats -- python tools/aux_rates.py 10 8   (200 x 4K BGR frames: fused colour embed embed_bgr_kernel<2, QM, false> = guarded / fast, <8, QM, true> = exact; extract_bgr_kernel)" profiles/${TAG}_kernel_stats_colour_n10.csv
grep '^{' $E/rocprof_stats.log | tail -1 > profiles/${TAG}_bench_under_rocprof.json
[ -f $E/rocprof_stats_colour.log ] && { echo "# commit $REV: python tools/aux_rates.py 10 8 (200 x 4K frames; run under rocprofv3 --kernel-trace --stats, whose kernel table is ${TAG}_kernel_stats_colour_n10.csv)"; grep -E "GB/s|fused|--" $E/rocprof_stats_colour.log | grep -v amdgpu; } > profiles/${ROUND}_aux_kernel_rates.txt
if [ -f $E/hbm_traffic.json ]; then
  sed "s/commit , tag/commit $REV, tag/" $E/hbm_traffic.json > profiles/hbm_traffic.json
  sed "s/commit , tag/commit $REV, tag/" $E/${ROUND}_pmc_summary.json > profiles/${ROUND}_pmc_summary.json
fi
for n in 3 10 63; do [ -f $E/sq_counters_n$n.txt ] && cp $E/sq_counters_n$n.txt profiles/${ROUND}_sq_counters_n$n.txt; done
[ -f $E/other_configs_bench.jsonl ] && cp $E/other_configs_bench.jsonl profiles/${ROUND}_other_configs_bench.jsonl
if ls $E/guarded_probe_n*.txt >/dev/null 2>&1; then
  { for n in 3 7 10 15; do [ -f $E/guarded_probe_n$n.txt ] && grep -v amdgpu.ids $E/guarded_probe_n$n.txt; done; } > profiles/${ROUND}_guarded_probe.txt
fi
if [ -f $E/tie_fallback_new.txt ]; then
  { echo "== this build ($REV)"; grep -v amdgpu.ids $E/tie_fallback_new.txt; echo; echo "== round-2 library (lib/variants/libsvsdct_r02.so, rebuilt from 68f741a), same box"; grep -v amdgpu.ids $E/tie_fallback_r02.txt; } > profiles/${ROUND}_tie_fallback_rate.txt
fi
[ -f $E/ab_vs_r02.txt ] && { echo "# commit $REV"; cat $E/ab_vs_r02.txt; } > profiles/${ROUND}_ab_vs_r02.txt
[ -f $E/ab_vs_earlier.txt ] && { echo "# commit $REV: tools/ab_bench.py, one process, 11 interleaved rounds, single launches between synchronisations; lib/variants/libsvsdct_r05.so and _r02.so rebuilt from git by make -C csrc r05 r02"; cat $E/ab_vs_earlier.txt; } > profiles/${ROUND}_ab_vs_earlier.txt
[ -f $E/placements.txt ] && { echo "# commit $REV: tools/placement_ab.py --pairs 6 --rounds 3 (sustained bursts of 8 launches; this = the product library, r05 / r02 = the libraries of those rounds rebuilt from git, this_copy = this build's launch with an empty payload: its access pattern with the arithmetic skipped, copy = 16 bytes per lane, linear)"; grep -v amdgpu.ids $E/placements.txt; } > profiles/${ROUND}_placements.txt
[ -f $E/ab_n15_n16.txt ] && { echo "# commit $REV: tools/ab_bench.py, 600 x 4K, single launches between synchronisations, 7 interleaved rounds - the two-row streaming kernel at its last n against the lane-per-block pocketfft kernel at the first n it has to take"; cat $E/ab_n15_n16.txt; } > profiles/${ROUND}_ab_n15_n16.txt
for i in 2 3; do [ -s $E/bench_process_$i.json ] && cp $E/bench_process_$i.json profiles/${TAG}_bench_process_$i.json; done
[ -f $E/parity_report.json ] && cp $E/parity_report.json profiles/${ROUND}_parity_report.json
[ -f $E/pipeline_overlap.json ] && cp $E/pipeline_overlap.json profiles/${ROUND}_pipeline_overlap.json
echo "profiles/${TAG}_* written at $REV"
