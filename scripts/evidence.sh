#!/bin/bash
# The round's evidence in one GPU call: everything lands in gpurun_out/ev/, from where the files are copied to profiles/<round>_*.
#   gpurun --timeout 2400 -- 'bash scripts/evidence.sh r06'
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOTD=$PWD
R=${1:-r06}
E=gpurun_out/ev; mkdir -p $E
export TMPDIR=/tmp
t() { timeout "$1" bash -c "$2" < /dev/null; }
echo "== GPU suite"; t 900 "python -m pytest tests -m gpu -q -p no:cacheprovider" > $E/${R}_gpu_suite.txt 2>&1; tail -3 $E/${R}_gpu_suite.txt
echo "== smoke"; t 300 "python __graft_entry__.py smoke" > $E/${R}_smoke.txt 2>&1; tail -3 $E/${R}_smoke.txt
echo "== bench (driver's arguments), twice"
t 600 "python bench.py --gpus 1 --steps 20 --warmup 5 2>$E/${R}_bench_err.log" > $E/${R}_bench_line_driver_args.json; cut -c1-260 $E/${R}_bench_line_driver_args.json
t 600 "python bench.py 2>>$E/${R}_bench_err.log" > $E/${R}_bench_line.json; cut -c1-260 $E/${R}_bench_line.json
t 300 "python bench.py --config 3 --no-dp-dry-run 2>>$E/${R}_bench_err.log" > $E/${R}_bench_config3.json; cut -c1-200 $E/${R}_bench_config3.json
t 300 "python bench.py --config 5 --no-dp-dry-run 2>>$E/${R}_bench_err.log" > $E/${R}_bench_config5.json; cut -c1-200 $E/${R}_bench_config5.json
echo "== traced step"; bash scripts/step_trace.sh ${R}ev > /dev/null 2>&1
for f in eager_breakdown eager_timeline small_kernel_chains; do mv gpurun_out/${R}ev_$f.txt $E/${R}_$f.txt; done; head -3 $E/${R}_eager_breakdown.txt | cut -c1-300
echo "== rocprofv3 --kernel-trace --stats of the bench command"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof_stats" -o st -- bash -c "cd $ROOTD && exec python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --no-dp-dry-run --no-reference-order --no-sustained" > "$ROOTD/$E/${R}_rocprof_stats_bench_line.json" 2>/dev/null < /dev/null)
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $E/${R}_rocprof_kernel_stats.csv
g=$(find gpurun_out/prof_stats -name "*kernel_trace.csv" | head -1); [ -n "$g" ] && MINL=15 python scripts/trace_by_grid.py "$g" igemm_tng wino3 skinny igemm_nng wino_gemm_g > $E/${R}_roofline_launch_durations.txt
rm -rf gpurun_out/prof_stats; head -8 $E/${R}_rocprof_kernel_stats.csv | cut -c1-160
echo "== laboratories"
{ t 120 "tools/wino_lab 128 32 20"; t 120 "tools/wino_lab 384 16 20"; t 120 "tools/wino_lab 256 32 20"; } > $E/${R}_wino_lab.txt 2>&1; grep -c "us  executed" $E/${R}_wino_lab.txt
echo "== skinny layers alone (kernel trace)"
{ for n in 128 64 256; do echo "# batch $n"; MINL=5 bash scripts/ktrace.sh sk skinny,reduce_small -- python scripts/kbench.py $n --only gconv4,gconv4y,dconv1; echo "# batch $n, round 1's VALU kernels (CG_SKINNY=2)"; CG_SKINNY=2 MINL=5 bash scripts/ktrace.sh sk skinny,reduce_small -- python scripts/kbench.py $n --only gconv4,gconv4y,dconv1; done; } > $E/${R}_skinny.txt 2>&1
grep -c skinny $E/${R}_skinny.txt
echo "== per-layer table"; t 300 "python scripts/kbench.py 128" > $E/${R}_kbench_per_layer.txt 2>&1; tail -2 $E/${R}_kbench_per_layer.txt
echo "== same-box switches (ms per step, images/s)"
{ STEPS=50 bash scripts/gpu_ab_env.sh CG_WINO3=1 CG_WINO3=0; STEPS=50 bash scripts/gpu_ab_env.sh CG_SKINNY=1 CG_SKINNY=2; STEPS=50 bash scripts/gpu_ab_env.sh CG_CONCURRENT_G_BOTH=1 CG_CONCURRENT_G_BOTH=0; } > $E/${R}_switches.txt 2>&1; cat $E/${R}_switches.txt
echo "== PMC"; bash scripts/pmc_kernels.sh $R > $E/${R}_pmc_log.txt 2>&1; cp gpurun_out/pmc/${R}_pmc_kernels.json $E/ 2>/dev/null; tail -12 $E/${R}_pmc_log.txt | cut -c1-200
rm -rf gpurun_out/pmc
