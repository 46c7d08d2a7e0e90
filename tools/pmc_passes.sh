#!/bin/bash
# PMC passes for the trace-and-shade kernel (one rocprofv3 run per counter set, never combined with
# tracing domains other than --kernel-trace).  usage: tools/pmc_passes.sh <tag> [workload]
# Results: gpurun_out/pmc_<tag>_<set>/ ; summarise with tools/rocpd_summary.py.
set -e
tag=$1; wl=${2:-S4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${tag}_${name} -o $wl -- python3 bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --kernel-only --workload $wl > gpurun_out/pmc_${tag}_${name}.log 2>&1
}
# (GRBM_GUI_ACTIVE rides in the SAME pass as the counters it is divided into: kernel times differ by 10 % from pass to pass)
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
# lane utilisation inside issued vector instructions: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), one pass
run sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetchsz FETCH_SIZE
run write WRITE_SIZE
