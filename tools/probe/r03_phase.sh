cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
UVO_PNP_SPEC=0 UVO_DBG_PHASE=1 timeout -k 10 120 python tools/prof_stereo.py 6 2>&1 | grep -v amdgpu | tail -8
