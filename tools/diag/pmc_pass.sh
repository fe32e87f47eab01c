# usage (GPU box, repo root): bash tools/diag/pmc_pass.sh TAG "COUNTER COUNTER ..."   (<= 8 SQ counters per pass)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pmc_$TAG -o pmc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-wide > gpurun_out/pmc_$TAG.log 2>&1
python3 tools/diag/pmc_summary.py gpurun_out/pmc_$TAG
