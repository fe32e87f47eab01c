cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ppl_stats -o s -- python3 tools/diag/time_fast_kernels.py > gpurun_out/ppl_stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/ppl_pmc -o pmc -- python3 tools/diag/time_fast_kernels.py > gpurun_out/ppl_pmc.log 2>&1
grep "us per call" gpurun_out/ppl_stats.log
grep "segment_kernel\|chain" gpurun_out/ppl_stats/s_kernel_stats.csv | cut -c1-200
