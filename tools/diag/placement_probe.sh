# usage (on the GPU box): bash tools/diag/placement_probe.sh LIB_A LIB_B
# Two builds of the library whose flagged-utterance kernel (ctc_exact_kernel, the tail of every fast-path call) differs only in
# where its code lies: duration of the kernel on the headline call with NOTHING flagged, and its instruction-fetch counters.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/placement
python3 tools/diag/ab_time.py "$@" 2>&1 | grep -v amdgpu | tee gpurun_out/placement/ab_time.txt
for lib in "$@"; do
  v=$(basename $lib .so)
  export E2E_CTC_LIB=$GRAFT_REPO_ROOT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/placement/stats_$v -o k -- python3 tools/diag/time_shape.py 256 1000 29 200 > gpurun_out/placement/stats_$v.log 2>&1
  rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/placement/pmc_sq_$v -o pmc -- python3 tools/diag/time_shape.py 256 1000 29 200 > gpurun_out/placement/pmc_sq_$v.log 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d gpurun_out/placement/pmc_sqc_$v -o pmc -- python3 tools/diag/time_shape.py 256 1000 29 200 > gpurun_out/placement/pmc_sqc_$v.log 2>&1
  echo "== $v"
  grep "ctc_exact_kernel" gpurun_out/placement/stats_$v/k_kernel_stats.csv | cut -d, -f2-4,6-7
  python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq_$v", "pmc_sqc_$v"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/placement/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctc_exact_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg): print("  %-28s %12.0f (n=%d)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
done
