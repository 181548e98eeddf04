set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_mini.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r04/t2_mini.log
timeout 900 python -m pytest tests/test_hip_pair.py tests/test_hip_gemm.py tests/test_hip_parity.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r04/t2_rest.log
PARITY_MODES="default:-1:-1" timeout 1500 python tools/parity_all_runs.py > gpurun_out/r04/parity2.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err
tail -40 gpurun_out/r04/t2_mini.log gpurun_out/r04/t2_rest.log gpurun_out/r04/parity2.log
