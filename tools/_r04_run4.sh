set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/_r04_dbg.py > gpurun_out/r04/dbg4.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_configs.py::test_baseline_config3_three_reference_runs_other_weights_noise_and_labels 2>&1 | tail -60 > gpurun_out/r04/t4_all.log
cat gpurun_out/r04/dbg4.log; tail -40 gpurun_out/r04/t4_all.log
