set -x
mkdir -p gpurun_out/r04
PARITY_MODES="default:-1:-1,x pairs + W:1:2" timeout 900 python tools/parity_all_runs.py sample_full10_16_nocfg sample_full10_16_nocfg_s2 sample_full10_16_nocfg_s3 > gpurun_out/r04/xw_parity.log 2>&1
tail -12 gpurun_out/r04/xw_parity.log
timeout 600 python -m pytest tests/test_hip_mini.py tests/test_hip_gemm.py -x -q -m gpu 2>&1 | tail -3
