# the 20-step form many times with UVO_TRACE: keep the trace of any run below 3300 pairs/s
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq 1 40); do
  v=$(UVO_TRACE=gpurun_out/tr_$i.csv python3 bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | tail -1 | python3 -c "import sys,json; print(int(json.loads(sys.stdin.readline())['value']))")
  echo "run $i: $v"
  if [ "$v" -gt 3300 ]; then rm -f gpurun_out/tr_$i.csv; fi
done
