set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_hip_mini.py tests/test_hip_parity.py tests/test_hip_pair.py tests/test_hip_variants.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r04/t7.log
timeout 1200 python -m pytest tests/test_hip_configs.py -m gpu -q -k "variants_full_width or trained_like or config1" 2>&1 | tail -30 >> gpurun_out/r04/t7.log
CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 64 > gpurun_out/r04/cfgbench7.log 2>&1
MASKBIT_AMD_NO_HALF_TILES=1 CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 > gpurun_out/r04/cfgbench7_nohalf.log 2>&1
cat gpurun_out/r04/t7.log; cat gpurun_out/r04/cfgbench7.log gpurun_out/r04/cfgbench7_nohalf.log
