cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/probe/ab_env.py - UVO_MAX_B=3 UVO_A_OVERLAP=2 - 2>/dev/null || exit 1
