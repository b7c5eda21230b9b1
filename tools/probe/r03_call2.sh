# round 3, GPU call 2: cold-start of the driver's 20-step form (lane priming), priority A/B in separate processes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_c2_b20_$i.json 2> gpurun_out/r03_c2_b20_$i.err || exit 1
  python -c "import json,sys; d=json.loads(open('gpurun_out/r03_c2_b20_$i.json').read().strip().splitlines()[-1]); print('20/5:', d['value'], d['step_latency_ms']['median'])"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 40 --no-cpu-baseline > gpurun_out/r03_c2_b20_w40.json 2> gpurun_out/r03_c2_b20_w40.err || exit 1
python -c "import json,sys; d=json.loads(open('gpurun_out/r03_c2_b20_w40.json').read().strip().splitlines()[-1]); print('20/40:', d['value'])"
for v in - UVO_PNP_PRIORITY=0 UVO_WORKER_WAIT=block-all UVO_WORKER_WAIT=spin; do
  timeout -k 10 200 python tools/probe/ab_env.py $v >> gpurun_out/r03_c2_ab.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/r03_c2_ab.log
