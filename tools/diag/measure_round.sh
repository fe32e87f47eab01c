set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_r01_v7.json 2> gpurun_out/bench_r01_v7.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v7 -o v4 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode > gpurun_out/prof_v7.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_v7_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/pmc_v7_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_v7_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/pmc_v7_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_v7_sq -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode > gpurun_out/pmc_v7_sq.log 2>&1
ls -R gpurun_out/prof_v7 gpurun_out/pmc_v7_fetch | head -20
cat gpurun_out/bench_r01_v7.json
