set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
CONFIG_BENCH_ONLY="configs[1]" timeout 600 python tools/config_bench.py 16 64 > gpurun_out/r04/cfgbench6.log 2>&1
PARITY_MODES="default:-1:-1,act3 pairs:3:0,single fp16:0:0" timeout 1500 python tools/parity_all_runs.py sample_full12_64_prenorm sample_full12_64_seq1024 sample_full12_64_outlier sample_full10_16_nocfg_outlier > gpurun_out/r04/parity6.log 2>&1
cat gpurun_out/r04/cfgbench6.log; tail -20 gpurun_out/r04/parity6.log
