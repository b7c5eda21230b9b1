#!/bin/bash
# the descriptor launch under different workgroup orders / grid sizes (UVO_DESC_ORDER, UVO_DESC_GRID, UVO_DESC_NBIG; surf.hip):
#   gpurun -- tools/probe/desc_order.sh "order grid nbig" ...
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  set -- $cfg
  UVO_DESC_ORDER=$1 UVO_DESC_GRID=$2 UVO_DESC_NBIG=$3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_do -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_do.log 2>&1 || exit 1
  echo "order $1 grid $2 nbig $3: $(python tools/probe/kstats.py prof_do 30 | grep -E 'descriptor64\(' | awk '{print $NF}') us"
  rm -rf gpurun_out/prof_do
done
