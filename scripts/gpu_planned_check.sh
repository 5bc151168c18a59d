cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_abi_step.py > gpurun_out/planned_pytest.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/planned_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/planned_bench.json 2> gpurun_out/planned_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/planned_bench.json; tail -5 gpurun_out/planned_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-graph > gpurun_out/planned_bench_eager.json 2>> gpurun_out/planned_bench.err; cut -c1-300 gpurun_out/planned_bench_eager.json
CG_PLANNED=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline > gpurun_out/legacy_bench.json 2>> gpurun_out/planned_bench.err; cut -c1-300 gpurun_out/legacy_bench.json
