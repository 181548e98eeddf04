set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_mini.py tests/test_hip_pair.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r04/t9.log
GEMM_AB_MINI=0 python tools/gemm_ab.py > gpurun_out/r04/ab9.log 2>&1
GEMM_AB_MINI=1 python tools/gemm_ab.py >> gpurun_out/r04/ab9.log 2>&1
GEMM_AB_MINI=0 python tools/gemm_ab.py >> gpurun_out/r04/ab9.log 2>&1
cat gpurun_out/r04/t9.log gpurun_out/r04/ab9.log
