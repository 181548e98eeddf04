set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r04/t10_all.log
PARITY_MODES="default:-1:-1" timeout 1800 python tools/parity_all_runs.py > gpurun_out/r04/parity10.log 2>&1
bash tools/profile_round4.sh r04 > gpurun_out/r04/profile10.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench10.json 2> gpurun_out/r04/bench10.err
tail -8 gpurun_out/r04/t10_all.log; grep "==" gpurun_out/r04/parity10.log; head -c 300 gpurun_out/r04/bench10.json
