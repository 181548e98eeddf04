set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
PARITY_MODES="default:-1:-1" timeout 900 python tools/parity_all_runs.py sample_full10_16_nocfg sample_full10_16_nocfg_s2 sample_full10_16_nocfg_s3 sample_full12_64_prenorm > gpurun_out/r04/parity8.log 2>&1
bash tools/profile_round4.sh r04 > gpurun_out/r04/profile8.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench8.json 2> gpurun_out/r04/bench8.err
tail -12 gpurun_out/r04/parity8.log; tail -5 gpurun_out/r04/profile8.log; head -c 300 gpurun_out/r04/bench8.json
