#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 1 ]; then export UVO_TAIL_SPLIT=1; else unset UVO_TAIL_SPLIT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ts -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_ts.log 2>&1 || exit 1
  echo "split=$v: $(python tools/probe/kstats.py prof_ts 30 | grep -E 'stereo_tail|extract3d_b' | awk '{print $1, $NF}' | tr '\n' ' ')"
  rm -rf gpurun_out/prof_ts
done
