cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_c3_b20_$i.json 2> gpurun_out/r03_c3_b20_$i.err || exit 1
  python -c "import json,sys; d=json.loads(open('gpurun_out/r03_c3_b20_$i.json').read().strip().splitlines()[-1]); print('20/5:', d['value'], d['roofline']['peak_measured'])"
done
