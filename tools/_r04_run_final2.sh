set -x
mkdir -p gpurun_out/r04
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench12.json 2> gpurun_out/r04/bench12.err
tail -c 1200 gpurun_out/r04/bench12.json
timeout 2400 bash tools/profile_round4.sh r04f > gpurun_out/r04/profile_r04f.log 2>&1
tail -3 gpurun_out/r04/profile_r04f.log
ls gpurun_out/prof_r04f/
