# usage: bash tools/probe/ab.sh "ENV1=.. ENV2=.." "ENV.." ...   -- interleaved repetitions of bench.py per configuration
for rep in 1 2 3; do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['step_latency_ms']['median'])"
  done
done
