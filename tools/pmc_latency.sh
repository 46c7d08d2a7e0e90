#!/bin/bash
# latency-oriented PMC passes (memory instruction levels, L1 traffic). usage: tools/pmc_latency.sh <tag> [workload]
set -e
tag=$1; wl=${2:-S4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${tag}_${name} -o $wl -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 --workload $wl > gpurun_out/pmc_${tag}_${name}.log 2>&1
}
run lvl SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_IFETCH_LEVEL
run tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
