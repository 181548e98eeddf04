set -x
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -q -m gpu -x -k "attention or pair or att" 2>&1 | tail -5
timeout 300 python tools/att_trace.py run 64 notrace > gpurun_out/r04/att_ab.log 2>&1; cat gpurun_out/r04/att_ab.log | tail -12
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_full64.py -q -m gpu -x 2>&1 | tail -5
PARITY_MODES="default:-1:-1" timeout 900 python tools/parity_all_runs.py sample_full12_64 sample_full12_64_s2 sample_full12_64_s3 sample_full10_16_nocfg sample_full10_16_nocfg_s2 sample_full10_16_nocfg_s3 > gpurun_out/r04/parity12.log 2>&1; grep "==" gpurun_out/r04/parity12.log
