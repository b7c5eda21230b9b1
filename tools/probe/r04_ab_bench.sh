#!/bin/bash
# pipelined throughput, previous build (ergo_uvo_amd/lib_ab/libuvo_hip_old.so) against the current one, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export UVO_HIP_LIB=$GRAFT_REPO_ROOT/ergo_uvo_amd/lib_ab/libuvo_hip_old.so; else unset UVO_HIP_LIB; fi
    python bench.py --steps ${1:-300} --blocks 5 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['block_values'])"
  done
done
