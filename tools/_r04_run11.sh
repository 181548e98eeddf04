set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_hip_mini.py tests/test_hip_parity.py tests/test_hip_sampling.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r04/t11.log
for cs in 4 2 1; do echo "### MASKBIT_AMD_COL_SPLIT=$cs" >> gpurun_out/r04/cfgbench11.log; MASKBIT_AMD_COL_SPLIT=$cs CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 32 >> gpurun_out/r04/cfgbench11.log 2>&1; done
cat gpurun_out/r04/t11.log; grep -E "###|images/s|ffn_down|attn_out" gpurun_out/r04/cfgbench11.log
