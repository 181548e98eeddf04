set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k batch_invariance 2>&1 | tail -8 > gpurun_out/r04/t3.log
for tag in "" nophase nodma nomma noread; do
  echo "##### lib tag '$tag' HT_MINI=1" >> gpurun_out/r04/trace3.log
  HT_TAG=$tag HT_MINI=1 timeout 300 python tools/ht_trace.py run 1 >> gpurun_out/r04/trace3.log 2>&1
done
echo "##### base lib HT_MINI=0" >> gpurun_out/r04/trace3.log
HT_MINI=0 timeout 300 python tools/ht_trace.py run 1 >> gpurun_out/r04/trace3.log 2>&1
echo "##### base lib HT_MINI=2" >> gpurun_out/r04/trace3.log
HT_MINI=2 timeout 300 python tools/ht_trace.py run 1 >> gpurun_out/r04/trace3.log 2>&1
PARITY_MODES="default:-1:-1,act3 pairs:3:1,single fp16:0:0" timeout 900 python tools/parity_all_runs.py sample_full12_64_outlier sample_full10_16_nocfg_outlier sample_full10_16_nocfg sample_full10_16_nocfg_s2 > gpurun_out/r04/parity3.log 2>&1
tail -5 gpurun_out/r04/t3.log; tail -30 gpurun_out/r04/parity3.log
