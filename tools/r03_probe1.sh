#!/bin/bash
# round 3, first GPU call: baselines + three cheap experiments
mkdir -p gpurun_out/r03
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
(timeout 400 python tools/cumask_probe.py 8 > gpurun_out/r03/cumask.log 2>&1; echo "rc=$?" >> gpurun_out/r03/cumask.log)
(timeout 300 python tools/batch_sweep.py 16 32 48 64 > gpurun_out/r03/batch_sweep.log 2>&1)
(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_base.json 2> gpurun_out/r03/bench_base.err)
(MASKBIT_AMD_WFROM=12 PM_MODES=3 timeout 500 python tests/diag/parity_modes.py > gpurun_out/r03/parity_wfrom12.log 2>&1)
(MASKBIT_AMD_WFROM=12 timeout 300 python bench.py --steps 2 --warmup 1 --mode wcorr --no-cpu-baseline --no-modes > gpurun_out/r03/bench_wfrom12.json 2> gpurun_out/r03/bench_wfrom12.err)
tail -n 40 gpurun_out/r03/cumask.log gpurun_out/r03/batch_sweep.log gpurun_out/r03/parity_wfrom12.log
python - <<'PY'
import json
for f in ("bench_base", "bench_wfrom12"):
    try:
        j = json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], {k: round(v["avg_us"], 1) for k, v in j["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
