# usage (on the GPU box, from the repo root): bash tools/diag/measure_round6.sh TAG [full]
# bench line + rocprofv3 kernel stats + SQ counters (+ FETCH/WRITE passes with PMC=1) of the default bench command
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-decode --no-wide"
[ "$2" = full ] && python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06_$TAG.json 2> gpurun_out/bench_r06_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o c2 -- python3 bench.py --steps 20 --warmup 5 $FLAGS > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_sq -o pmc -- python3 bench.py --steps 5 --warmup 2 $FLAGS > gpurun_out/pmc_${TAG}_sq.log 2>&1
python3 tools/diag/pmc_summary.py gpurun_out/pmc_${TAG}_sq
if [ -n "$PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o pmc -- python3 bench.py --steps 5 --warmup 2 $FLAGS > gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_write -o pmc -- python3 bench.py --steps 5 --warmup 2 $FLAGS > gpurun_out/pmc_${TAG}_write.log 2>&1
python3 tools/diag/pmc_summary.py gpurun_out/pmc_${TAG}_fetch
python3 tools/diag/pmc_summary.py gpurun_out/pmc_${TAG}_write
fi
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-180
grep '^{"metric' gpurun_out/prof_$TAG.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','module_ms_per_step')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['peak_measured'])
"
if [ "$2" = full ]; then
# the other kernel families, the wide-alphabet share in both dtypes, the word-piece shapes
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_others_$TAG -o others -- python3 tools/diag/profile_others.py > gpurun_out/prof_others_$TAG.log 2>&1 < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5_f32_$TAG -o c5 -- python3 tools/diag/profile_c5.py > gpurun_out/prof_c5_f32_$TAG.log 2>&1 < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5_bf16_$TAG -o c5 -- python3 tools/diag/profile_c5.py bf16 > gpurun_out/prof_c5_bf16_$TAG.log 2>&1 < /dev/null
bash tools/diag/profile_shape.sh wp8k_$TAG 64 256 8000 200 > /dev/null
bash tools/diag/profile_shape.sh wp32k_$TAG 16 150 32000 120 > /dev/null
bash tools/diag/profile_shape.sh mid_$TAG 256 1000 200 200 > /dev/null
fi
if [ "$2" = full ]; then
# round 6: the emission regimes with the flagged launch's phases, the cliff scan, the beam's phases
python3 tools/diag/flagged_phases.py > gpurun_out/r06_flagged_phases_$TAG.txt 2>&1
python3 tools/diag/cliff_scan.py > gpurun_out/r06_cliff_scan_$TAG.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_regimes_$TAG -o regimes -- python3 tools/diag/emission_regimes.py > gpurun_out/prof_regimes_$TAG.log 2>&1 < /dev/null
fi
