#!/bin/bash
# tools/probe/gpu.sh -- the measurement steps behind DESIGN.md's numbers, one script instead of a file per GPU call:
#   gpurun -- 'tools/probe/gpu.sh <step> [args] [-- <step> [args]] ...'
# Steps (outputs under gpurun_out/; a failing step stops the chain):
#   tests [pytest args]         python -m pytest <args, default: tests -m gpu> -x -q            -> <tag>_tests.log
#   bench [bench.py args]       one bench.py line (stdout)                                       
#   bench20 [n] [bench args]    the driver's form n times (--steps 20 --warmup 5 + args, e.g. --no-trace), value + blocks
#   ab <steps>                  previous build (ergo_uvo_amd/lib_ab/libuvo_hip_old.so) against the current one, interleaved
#   sweep <steps>               overlap / depth / PnP-slot settings of the one-pair pipeline
#   batch <steps>               one- and two-pair launch sets over depth / overlap settings
#   prof <tag> [sync|pipe|batch2]  rocprofv3 --kernel-trace --stats of a synchronous run / the pipelined bench / two-pair launch sets
#   hess-split <tag>            per-octave detection kernels (UVO_HESSIAN_SPLIT=1) and the merged launch, old build and new
#   pmc <tag> <kernel pattern> [split]   counter passes (separate --pmc runs: SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE) of a synchronous run
#   ctx-reuse [pairs]           five contexts in a row, stream pool on and off
#   sens [steps]                a kernel of known duration added to stage A (one wave / chip-filling): how the cadence follows thin and fat time
#   stamps                      phases of the detection launch's workgroups (UVO_HESS_STAMPS -> tools/probe/hess_stamps.py)
#   rates                       issue-rate probe (built by hipcc here if missing)
#   topo                        what the rank pinning reads on this box
#   configs                     tools/bench_configs.py
#   mono <tag>                  rocprofv3 kernel stats of the mono loop at C4 (tools/prof_mono.py contract | 1px), + pipelined rate (tools/probe/mono_pipe.py)
#   sift <tag>                  the SIFT detector: timing, kernel stats, the stereo loop on SIFT under the profiler
#   sift-pmc                    counter passes of the SIFT detector (one per group) -> tools/pmc_summary_sift.py
#   binary <tag>                the AKAZE and ORB detectors: timing (CPU oracle beside it), kernel stats of each
#   budget [host_budget args]   tools/host_budget.py: {spin, sleep, block-all, device} x {2, 4, 16 CPUs}, both bench forms
#   final <tag>                 the closing pass of a round: bench (600 steps and the driver's form), kernel stats, counters, configs
# TAG=<name> prefixes the output files (default r05).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
mkdir -p gpurun_out
TAG=${TAG:-r05}
OLD=$PWD/ergo_uvo_amd/lib_ab/libuvo_hip_old.so
line() { python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % os.environ.get('LBL',''), d['value'], d.get('block_values'))"; }
brun() { python bench.py --blocks 5 --timed-only "$@" 2>/dev/null | LBL="${LBL:-$*}" line; }
step() {
  s=$1; shift
  case $s in
    tests)   if [ $# -eq 0 ]; then set -- tests -m gpu; fi
             timeout -k 10 1100 python -m pytest "$@" -x -q > gpurun_out/${TAG}_tests.log 2>&1; rc=$?; tail -4 gpurun_out/${TAG}_tests.log; return $rc ;;
    bench)   python bench.py "$@" ;;
    bench20) n=${1:-3}; shift; for i in $(seq $n); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20/5:', d['value'], 'first', d['value_first_block'], d['block_values'], 'gap max', d['collect_gap_ms']['max'], d['collect_gap_ms']['argmax'])" || return 1; done ;;
    ab)      for rep in 1 2; do UVO_HIP_LIB=$OLD LBL=old brun --steps ${1:-300} || return 1; LBL=new brun --steps ${1:-300} || return 1; done ;;
    sweep)   n=${1:-300}
             LBL=default brun --steps $n --depth 6; UVO_A_OVERLAP=3 LBL="A_OVERLAP=3 depth 6" brun --steps $n --depth 6; UVO_A_OVERLAP=3 LBL="A_OVERLAP=3 depth 8" brun --steps $n --depth 8
             UVO_MAX_B=4 LBL="MAX_B=4" brun --steps $n --depth 6; UVO_MAX_B=2 LBL="MAX_B=2" brun --steps $n --depth 6
             LBL="depth 8" brun --steps $n --depth 8; LBL="depth 7" brun --steps $n --depth 7; LBL=default brun --steps $n --depth 6 ;;
    batch)   n=${1:-300}
             LBL="batch 1 depth 6" brun --steps $n --batch 1 --depth 6; LBL="batch 2 depth 6" brun --steps $n --batch 2 --depth 6; LBL="batch 2 depth 8" brun --steps $n --batch 2 --depth 8
             UVO_A_OVERLAP2=1 LBL="batch 2 depth 6 A_OVERLAP2=1" brun --steps $n --batch 2 --depth 6; UVO_A_OVERLAP2=1 LBL="batch 2 depth 4 A_OVERLAP2=1" brun --steps $n --batch 2 --depth 4
             LBL="batch 1 depth 6" brun --steps $n --batch 1 --depth 6 ;;
    prof)    t=$1; kind=${2:-sync}
             case $kind in
               sync)   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_${t}_sync.log 2>&1 || return 1; python tools/probe/kstats.py prof_${t}_sync 26 ;;
               pipe)   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_pipe -- python3 bench.py --steps 200 --blocks 1 --timed-only > gpurun_out/prof_${t}_pipe.log 2>&1 || return 1; python tools/probe/kstats.py prof_${t}_pipe 26 ;;
               batch2) UVO_A_OVERLAP2=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_batch2 -- python3 bench.py --steps 60 --blocks 1 --timed-only --batch 2 > gpurun_out/prof_${t}_batch2.log 2>&1 || return 1; python tools/probe/kstats.py prof_${t}_batch2 26 ;;
             esac ;;
    hess-split) t=$1
             for v in old new; do
               if [ $v = old ]; then export UVO_HIP_LIB=$OLD; else unset UVO_HIP_LIB; fi
               UVO_HESSIAN_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_${v}_split -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_${t}_${v}_split.log 2>&1 || return 1
               rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_${v}_merged -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_${t}_${v}_merged.log 2>&1 || return 1
               echo "== $v, one launch per octave"; python tools/probe/kstats.py prof_${t}_${v}_split 30 | grep -i hessian
               echo "== $v, merged launch"; python tools/probe/kstats.py prof_${t}_${v}_merged 30 | grep -i hessian
             done; unset UVO_HIP_LIB ;;
    pmc)     t=$1; pat=$2; [ "$3" = split ] && export UVO_HESSIAN_SPLIT=1
             rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_${t}_sq1 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${t}_sq1.log 2>&1 || return 1
             rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_${t}_sq2 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${t}_sq2.log 2>&1 || return 1
             rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${t}_fetch -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${t}_fetch.log 2>&1 || return 1
             rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${t}_write -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${t}_write.log 2>&1 || return 1
             unset UVO_HESSIAN_SPLIT
             for p in sq1 sq2 fetch write; do python tools/probe/pmc_quick.py pmc_${t}_${p} "$pat" 2; done ;;
    ctx-reuse) python tools/probe/ctx_reuse.py ${1:-400} 2>/dev/null | grep pool; UVO_STREAM_POOL=0 python tools/probe/ctx_reuse.py ${1:-400} 2>/dev/null | grep pool ;;
    sens)    n=${1:-300}; LBL=base brun --steps $n || return 1
             for us in 20 40; do UVO_PROBE_THIN_US=$us LBL="one wave held +$us us per pair" brun --steps $n; UVO_PROBE_FAT_US=$us LBL="768 LDS-filling blocks +$us us" brun --steps $n; done
             LBL=base brun --steps $n ;;
    stamps)  # the measurement build (make STAMPS=1 BUILD=build_stamps OUT=../lib_ab/libuvo_hip_stamps.so in ergo_uvo_amd/csrc): stamps cost time when compiled in
             make -C ergo_uvo_amd/csrc -s -j8 STAMPS=1 BUILD=build_stamps OUT=../lib_ab/libuvo_hip_stamps.so || return 1
             UVO_HIP_LIB=$PWD/ergo_uvo_amd/lib_ab/libuvo_hip_stamps.so UVO_HESS_STAMPS=gpurun_out/${TAG}_hess_stamps.csv python tools/prof_stereo.py 8 > /dev/null 2>&1; python tools/probe/hess_stamps.py gpurun_out/${TAG}_hess_stamps.csv
             UVO_HIP_LIB=$PWD/ergo_uvo_amd/lib_ab/libuvo_hip_stamps.so UVO_DESC_STAMPS=gpurun_out/${TAG}_desc_stamps.csv python tools/prof_stereo.py 8 > /dev/null 2>&1; python tools/probe/desc_stamps.py gpurun_out/${TAG}_desc_stamps.csv ;;
    rates)   [ tools/probe/issue_rate_probe -nt tools/probe/issue_rate_probe.hip ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probe/issue_rate_probe.hip -o tools/probe/issue_rate_probe || return 1
             tools/probe/issue_rate_probe ;;
    topo)    for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; grep -E "simd_count|cpu_cores_count|location_id|domain|drm_render_minor" $n/properties 2>&1; done
             ls -la /dev/dri /dev/kfd 2>&1; nproc; cat /sys/devices/system/node/node*/cpulist 2>&1; env | grep -E "VISIBLE|ROCR|HIP_|GPU_"
             python3 -c "import sys; sys.path.insert(0, '.'); from ergo_uvo_amd import multirank; print(multirank.visible_gpus())" ;;
    configs) python tools/bench_configs.py > gpurun_out/bench_${TAG}_configs.json 2> gpurun_out/bench_${TAG}_configs.err || { tail -5 gpurun_out/bench_${TAG}_configs.err; return 1; }; cat gpurun_out/bench_${TAG}_configs.json ;;
    mono)    t=${1:-$TAG}
             for kind in contract 1px; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_mono_$kind -- python3 tools/prof_mono.py $kind 32 > gpurun_out/prof_${t}_mono_$kind.log 2>&1 || return 1; tail -1 gpurun_out/prof_${t}_mono_$kind.log; python tools/probe/kstats.py prof_${t}_mono_$kind 22; done
             python tools/probe/mono_pipe.py 6 2>/dev/null | tail -3; python tools/probe/mono_pipe.py 14 2>/dev/null | tail -3 ;;
    sift)    t=${1:-$TAG}
             python3 tools/prof_sift.py 20 --cpu > gpurun_out/${t}_sift_time.log 2>&1 || return 1; cat gpurun_out/${t}_sift_time.log
             rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_sift -- python3 tools/prof_sift.py 5 > gpurun_out/prof_${t}_sift.log 2>&1 || return 1
             rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_siftvo -- python3 tools/bench_configs.py SIFTVO > gpurun_out/prof_${t}_siftvo.log 2>&1 || return 1; tail -1 gpurun_out/prof_${t}_siftvo.log ;;
    sift-pmc) for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
               case "$g" in FETCH*) d=fetch ;; WRITE*) d=write ;; SQ_INSTS_VALU*) d=sq1 ;; *) d=sq2 ;; esac
               rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/pmc_sift_$d -- python3 tools/prof_sift.py 4 > gpurun_out/pmc_sift_$d.log 2>&1 || return 1
             done; python tools/pmc_summary_sift.py ;;
    binary)  t=${1:-$TAG}
             python3 tools/prof_binary.py both 20 --cpu > gpurun_out/${t}_binary_time.log 2>&1 || { tail -5 gpurun_out/${t}_binary_time.log; return 1; }; cat gpurun_out/${t}_binary_time.log
             for d in akaze orb; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${t}_$d -- python3 tools/prof_binary.py $d 5 > gpurun_out/prof_${t}_$d.log 2>&1 || return 1; echo "== $d"; python tools/probe/kstats.py prof_${t}_$d 24; done ;;
    budget)  python tools/host_budget.py --out gpurun_out/${TAG}_host_budget.json "$@" ;;
    final)   t=${1:-$TAG}
             python bench.py > gpurun_out/bench_${t}_final.json 2> gpurun_out/bench_${t}_final.err || return 1; tail -c 300 gpurun_out/bench_${t}_final.json; echo
             python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${t}_20steps.json 2> gpurun_out/bench_${t}_20steps.err || return 1
             python bench.py --gpus 1 --backend nccl --force-dist --steps 200 --no-cpu-baseline > gpurun_out/bench_${t}_1rank_rccl.json 2> gpurun_out/bench_${t}_1rank_rccl.err || return 1
             step prof $t sync || return 1; step prof $t pipe || return 1
             step pmc $t k_hessian_nms_all || return 1 ;;
    *) echo "unknown step $s"; return 2 ;;
  esac
}
args=()
for a in "$@" --; do
  if [ "$a" = -- ]; then
    if [ ${#args[@]} -gt 0 ]; then echo "### ${args[*]}"; step "${args[@]}" || { echo "step '${args[*]}' failed"; exit 1; }; fi
    args=()
  else args+=("$a"); fi
done
