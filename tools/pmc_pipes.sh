#!/bin/bash
# Pipe-utilisation counters of the inference kernels (fp32-storage forward), one rocprofv3 pass per counter group, --kernel-trace only:
#   bash tools/pmc_pipes.sh r02      ->  profiles/r02_fp32_pipe_counters.csv  (per kernel: averages per dispatch)
set -e
TAG=${1:-r02}
ROOTD=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTD/gpurun_out/${TAG}_pipes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--dtype fp32 --no-cpu-baseline --train-steps 0 --no-conditioning --no-v3 --no-48k --no-modes --steps 20 --warmup 3"
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); d=$OUT/g$i; mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $ROOTD/bench.py $ARGS > $d/log.txt 2>&1 || { tail -n 5 $d/log.txt; exit 1; }
  echo "done group $i"
done
cd $ROOTD
python3 - "$OUT" "profiles/${TAG}_fp32_pipe_counters.csv" <<'PY'
import csv, glob, sys, collections
out_dir, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in acc.values() for c in k})
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel", "Dispatches"] + names)
    for k, v in sorted(acc.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
        if not any(t in k for t in ("mrf_kernel", "odconv", "conv_out", "gen_prologue")):
            continue
        n = max(x[0] for x in v.values())
        w.writerow([k[:110], n] + [round(v[c][1] / v[c][0], 1) if c in v and v[c][0] else "" for c in names])
print("wrote", dst)
PY
mkdir -p gpurun_out/${TAG}_profiles && cp profiles/${TAG}_fp32_pipe_counters.csv gpurun_out/${TAG}_profiles/
rm -rf $OUT    # the raw csv files are large; the summary is what is kept
