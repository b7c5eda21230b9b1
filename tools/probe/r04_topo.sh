#!/bin/bash
# what the rank pinning reads (ergo_uvo_amd/multirank.py: visible_gpus) on this box
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; grep -E "simd_count|cpu_cores_count|location_id|domain|drm_render_minor" $n/properties 2>&1; done
ls -la /dev/dri /dev/kfd 2>&1
nproc; cat /sys/devices/system/node/node*/cpulist 2>&1; taskset -p $$
env | grep -E "VISIBLE|ROCR|HIP_|GPU_" 
python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from ergo_uvo_amd import multirank
print(multirank.visible_gpus())
PY
