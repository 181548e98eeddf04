set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_parity.py tests/test_hip_gemm.py tests/test_hip_pair.py tests/test_hip_variants.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/t1.log
PARITY_MODES="default:-1:-1" python tools/parity_all_runs.py > gpurun_out/r04/parity1.log 2>&1
tail -30 gpurun_out/r04/t1.log gpurun_out/r04/parity1.log
