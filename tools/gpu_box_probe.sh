#!/bin/bash
# what kind of box is this?  clocks / power state from rocm-smi, idle and while the headline step runs, next to the
# start-up latency of k_fir_blk (tools/fir_blk_timeline.py: prologue ~8 us on the faster boxes of the pool, ~20 us on the slower)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-box}
{
  echo "== idle"; rocm-smi --showclocks --showperflevel --showpower --showmemuse 2>&1 | grep -v "^$" | head -40
  rocm-smi --showclkfrq 2>&1 | grep -E "\*|Supported" | head -30
  echo "== partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -i "partition" | head -6; cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | tr "\n" " "; echo
  rocm-smi --showtopo 2>&1 | grep -i "numa" | head -4
  echo "== kernel params"; cat /proc/cmdline; cat /sys/module/amdgpu/parameters/noretry /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/sched_policy 2>&1 | tr '\n' ' '; echo
  cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1
  echo "== under load"
  (python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 20000 > "$O/${V}_bench_long.json" 2>&1 &)
  sleep 8
  rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -30
  wait
  sleep 6
  tail -1 "$O/${V}_bench_long.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
  echo "== timeline"; timeout 100 python tools/fir_blk_timeline.py 2>&1 | grep -E "launch|prologue|pair  0|pair  1:"
} 2>&1 | tee "$O/${V}_box_probe.txt"
