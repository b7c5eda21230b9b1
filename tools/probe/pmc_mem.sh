#!/bin/bash
# memory-pipeline counters (texture addresser, vector L1, L2) of the descriptor and survivor-finish launches: gpurun -- tools/probe/pmc_mem.sh
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
for set in "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c1-6)
  echo "== $set"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_mem_$n -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_mem_$n.log 2>&1 || { tail -3 gpurun_out/pmc_mem_$n.log; continue; }
  python tools/probe/pmc_quick.py pmc_mem_$n "k_descriptor64(" 2
  python tools/probe/pmc_quick.py pmc_mem_$n "k_hessian_finish" 2
done
