set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in S2 S3 S4 S5; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --workload $wl > gpurun_out/r1g_bench_$wl.json 2> gpurun_out/r1g_bench_$wl.err
done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r1g_prof -o S4 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload S4 > gpurun_out/r1g_prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/r1g_prof/*/S4_results.db > gpurun_out/r1g_S4_kernel_trace_stats.txt 2>&1 || python3 tools/rocpd_summary.py $(find gpurun_out/r1g_prof -name "*.db" | head -1) > gpurun_out/r1g_S4_kernel_trace_stats.txt
bash tools/pmc_passes.sh r1g S4
for s in sq sq2 tcc fetchsz write; do python3 tools/rocpd_summary.py $(find gpurun_out/pmc_r1g_$s -name "*.db" | head -1); done > gpurun_out/r1g_S4_pmc.txt
