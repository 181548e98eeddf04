#!/bin/bash
# rocprofv3 evidence (rounds 4-6) of the product default (guided forward with the weight-correction mini-tiles) on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag>
#   -> gpurun_out/prof_<tag>/{kt, pmc/<gemm>.<counter>, pmc/dec.<counter>}
# Kernel-trace statistics of the default bench workload, then separate --pmc passes (never combined with other trace domains) over the
# trunk GEMM shapes as the guided forward runs them (CFG pair tiles) and over the decoder.
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PAIR_ONE_MINI=1
cd /tmp
# (round 6: the bench keeps its own HIP-event kernel timing on under the tracer, so that ONE run carries both figures of the dominant kernel: the line's
#  roofline.avg_launch_us and the trace's average duration -- boxes sustain different clocks, two runs on two boxes do not compare)
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --no-traffic > $OUT/bench_under_rocprof.log 2>&1
if [ -n "${PROFILE_ONLY_KT:-}" ]; then
  cd $ROOT
  DB=$(find $OUT/kt -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB $OUT/kernel_trace_stats.md > /dev/null
  grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_under_rocprof.json"))
r = d["roofline"]
print("bench line under the tracer:", round(d["value"], 2), "images/s;", r["kernel"], "avg_launch_us (HIP events, every 8th forward)", round(r["avg_launch_us"], 1), "clock", d["telemetry"].get("effective_clock_mhz"))
PY
  head -5 $OUT/kernel_trace_stats.md | cut -c1-160
  exit 0
fi
for shape in qkv attn_out ffn_up ffn_down; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc/$shape.$ctr -o run -- python $ROOT/tools/pair_one.py $shape 3 > $OUT/pmc_$shape.$ctr.log 2>&1
  done
done
for grp in "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc/ffn_up.$name -o run -- python $ROOT/tools/pair_one.py ffn_up 3 > $OUT/pmc_ffn_up.$name.log 2>&1
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc/dec.$ctr -o run -- python $ROOT/tools/decode_one.py 64 3 > $OUT/pmc_dec.$ctr.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/kt_dec -o kt -- python $ROOT/tools/decode_one.py 64 3 > $OUT/dec_under_rocprof.log 2>&1
cd $ROOT
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $OUT/kernel_trace_stats.md > /dev/null
DB2=$(find $OUT/kt_dec -name "*.db" | head -1)
python tools/rocprof_summary.py $DB2 $OUT/decoder_kernel_trace_stats.md > /dev/null
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > /dev/null
python tools/pmc_decoder_summary.py $OUT/pmc 64 3 $OUT/decoder_pmc.md
tail -1 $OUT/bench_under_rocprof.log | cut -c1-200
