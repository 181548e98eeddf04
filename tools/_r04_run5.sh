set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/r04/t5_all.log
echo "##### packed W4: base lib HT_MINI=1" > gpurun_out/r04/trace5.log
HT_MINI=1 timeout 300 python tools/ht_trace.py run 1 >> gpurun_out/r04/trace5.log 2>&1
timeout 1500 python tools/wmask_parity.py 15 5 13 7 > gpurun_out/r04/wmask5.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r04/bench5.json 2> gpurun_out/r04/bench5.err
tail -60 gpurun_out/r04/t5_all.log; cat gpurun_out/r04/wmask5.log | tail -8
