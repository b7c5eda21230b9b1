cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 768 1024 1152 1536; do
UVO_MATCH_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03m_g$g -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03m_g$g.log 2>&1 || exit 1
done
