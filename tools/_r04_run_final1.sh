set -x
mkdir -p gpurun_out/r04
PARITY_MODES="default:-1:-1" timeout 1800 python tools/parity_all_runs.py > gpurun_out/r04/parity11.log 2>&1
grep "==" gpurun_out/r04/parity11.log
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04/suite11.log
cat gpurun_out/r04/suite11.log
timeout 600 python tools/config_bench.py 16 64 > gpurun_out/r04/config_bench11.log 2>&1
grep "images/s" gpurun_out/r04/config_bench11.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench11.json 2> gpurun_out/r04/bench11.err
tail -c 1500 gpurun_out/r04/bench11.json
