# usage (from the repo root, on the GPU box): bash tools/profile_round.sh TAG GIT_REV
# (before the call, where hipcc cross-compiles: tools/kres.sh > profiles/TAG_kres.txt — the spill table the counters file quotes)
# bench lines, a kernel trace and the PMC passes of S3 / S4 / S5 -> gpurun_out/<TAG>_*; copy what is to be judged into profiles/
set -e
tag=$1; rev=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in S2 S3 S4 S5; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --workload $wl --no-secondary --no-cpu-baseline > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err
done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o S4 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --workload S4 > gpurun_out/${tag}_prof.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/${tag}_prof -name "*.db" | head -1) > gpurun_out/${tag}_S4_kernel_trace_stats.txt
# the same with --kernel-only (whole frames through rtx_render_device, one stream: what roofline.achieved is measured on): every
# trace_shade dispatch of this trace is one frame with nothing beside it, its average is the bench's kernel_ms_avg
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_profk -o S4 -- python3 bench.py --no-cpu-baseline --no-secondary --kernel-only --steps 20 --warmup 3 --workload S4 > gpurun_out/${tag}_profk.log 2>&1
python3 tools/rocpd_summary.py $(find gpurun_out/${tag}_profk -name "*.db" | head -1) > gpurun_out/${tag}_S4_kernel_only_trace_stats.txt
for wl in S3 S4 S5; do
  bash tools/pmc_passes.sh ${tag}_$wl $wl
  for s in sq sq2 sq3 tcc fetchsz write; do python3 tools/rocpd_summary.py $(find gpurun_out/pmc_${tag}_${wl}_$s -name "*.db" | head -1); done > gpurun_out/${tag}_${wl}_pmc.txt
done
python3 tools/pmc_counters.py $tag $rev S3 S4 S5 > gpurun_out/${tag}_counters.json
# the raw rocprofv3 databases are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf gpurun_out/pmc_${tag}_* gpurun_out/${tag}_prof gpurun_out/${tag}_profk
