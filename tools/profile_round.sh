#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03
# Raw output goes to gpurun_out/<tag>_prof/ (scratch); tools/profile_summary.py turns it into profiles/ on either side.
# Counter passes are separate runs with --kernel-trace only (never mixed with other trace domains).
set -e
TAG=${1:-r03}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
INF="--no-cpu-baseline --train-steps 0 --no-conditioning --no-v3 --no-48k --no-modes --steps 40 --warmup 5"
run() { d=$OUT/$1; shift; mkdir -p $d; "$@" > $d/log.txt 2>&1 || { tail -n 5 $d/log.txt; exit 1; }; echo "done $d"; }
# 1. kernel-trace of the DEFAULT bench run (headline + modes + train + conditioning + 48k + plain V3)
run bench_trace rocprofv3 --kernel-trace --output-format csv -d $OUT/bench_trace -- python3 $ROOTD/bench.py --steps 100
grep '^{"metric"' $OUT/bench_trace/log.txt > $OUT/bench_line_profiled.json || true
# 2. HBM traffic, inference, per storage type: FETCH_SIZE and WRITE_SIZE in separate passes
# (mixed = the headline mode: fp16 storage through up1, fp32 storage behind it, MRF chain with two-product operands - summary tag fp32w16)
for dt in mixed bf16; do
  run fetch_$dt rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$dt -- python3 $ROOTD/bench.py --dtype $dt $INF
  run write_$dt rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$dt -- python3 $ROOTD/bench.py --dtype $dt $INF
done
# 3. HBM traffic, training step
TR="--no-cpu-baseline --train-steps 2 --train-warmup 1 --no-conditioning --no-v3 --no-48k --no-modes --steps 5 --warmup 1"
run fetch_train rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_train -- python3 $ROOTD/bench.py $TR
run write_train rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_train -- python3 $ROOTD/bench.py $TR
# 4. summaries (small; the raw csv files stay in gpurun_out/)
cd $ROOTD
python3 tools/profile_summary.py stats $OUT/bench_trace profiles/${TAG}_bench_kernel_stats.csv
python3 tools/profile_summary.py traffic $OUT/fetch_mixed $OUT/write_mixed profiles/${TAG}_mixed_pmc_traffic.json --dtype fp32w16 --workload "bench.py --dtype mixed $INF"
python3 tools/profile_summary.py traffic $OUT/fetch_bf16 $OUT/write_bf16 profiles/${TAG}_bf16_pmc_traffic.json --dtype bf16 --workload "bench.py --dtype bf16 $INF"
python3 tools/profile_summary.py traffic $OUT/fetch_train $OUT/write_train profiles/${TAG}_train_bf16_pmc_traffic.json --dtype train_bf16 --workload "bench.py $TR"
cp $OUT/bench_line_profiled.json profiles/${TAG}_bench_line_profiled.json 2>/dev/null || true
mkdir -p gpurun_out/${TAG}_profiles && cp profiles/${TAG}_* gpurun_out/${TAG}_profiles/
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
