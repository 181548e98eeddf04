set -x
mkdir -p gpurun_out/r04
CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 64 > gpurun_out/r04/xw_bench_default.log 2>&1
MASKBIT_AMD_ACT_SPLIT=1 CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 64 > gpurun_out/r04/xw_bench_act1.log 2>&1
grep -E "images/s|gemm|ln" gpurun_out/r04/xw_bench_default.log gpurun_out/r04/xw_bench_act1.log
