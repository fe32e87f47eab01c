# usage (on the GPU box): bash tools/diag/profile_shape.sh TAG B T V S -- rocprofv3 kernel stats of one loss call shape
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shape_$TAG -o s -- python3 tools/diag/time_shape.py "$@" > gpurun_out/shape_$TAG.log 2>&1 < /dev/null
grep "us per call" gpurun_out/shape_$TAG.log
f=$(find gpurun_out/shape_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -d, -f1-4 | cut -c1-200
