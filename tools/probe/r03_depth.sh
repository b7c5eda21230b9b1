cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 3 4 5 6; do
  for rep in 1 2 3; do
    python3 bench.py --steps 20 --warmup 5 --depth $d --timed-only 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('depth $d steps 20:', d['value'])"
  done
  python3 bench.py --steps 600 --warmup 20 --depth $d --timed-only 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('depth $d steps 600:', d['value'])"
done
