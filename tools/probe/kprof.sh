# per-kernel average times of a synchronous C3 run: bash tools/probe/kprof.sh <tag> [rows]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$1 -- python3 tools/prof_stereo.py 16 > gpurun_out/kprof_$1.log 2>&1
python3 tools/probe/kstats.py kprof_$1 ${2:-16}
