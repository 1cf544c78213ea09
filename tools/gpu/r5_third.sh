#!/bin/bash
# round 5, third GPU session: staging with the 4-8 MB chunk rule, ring on/off per direction, placement probe under counters
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 300 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || { tail -20 $O/pcie_rate.txt; exit 1; }
sed -n 8,22p $O/pcie_rate.txt
timeout -k 10 300 python tools/stage_chunk_sweep.py > $O/chunk_sweep.txt 2>&1 || { tail -20 $O/chunk_sweep.txt; exit 1; }
cat $O/chunk_sweep.txt
REPS=6
for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "GRBM_UTCL2_BUSY TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo "$pmc" | tr ' ' '+' | cut -c1-60)
  echo "== pmc $pmc" | tee -a $O/placement_counters.txt
  timeout -k 10 400 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/pc_$tag -- python tools/placement_probe.py --pairs 3 --reps $REPS --tag "$pmc" > $O/pc_$tag.log 2>&1
  rc=$?; [ $rc -eq 0 ] || { echo "rc=$rc"; tail -5 $O/pc_$tag.log; }
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout - stopping"; exit $rc; fi
  grep -E "pair|again" $O/pc_$tag.log >> $O/placement_counters.txt
  python tools/placement_counters.py $O/pc_$tag $REPS >> $O/placement_counters.txt 2>&1
  find $O/pc_$tag -name "*.csv" -size +2M -delete
done
cat $O/placement_counters.txt | cut -c1-230
