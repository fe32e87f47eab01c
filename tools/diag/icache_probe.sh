# usage (on the GPU box): bash tools/diag/icache_probe.sh
# The code-placement effect of DESIGN.md 4.1 (the same ISA 6.5x slower after a neighbouring kernel's growth moved it): time the
# regular library against a build WITHOUT the 64 KB alignment of the fast-path kernels (build/diag/ab_noalign.so:
# tools/diag/build_variant.sh noalign "-DE2E_KERNEL_ALIGN=" ctc_loss_fast.hip ctc_loss_fast_h1.hip), and if they differ, collect the
# instruction-fetch counters of both (separate --pmc passes, kernel stats in their own run).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/icache
python3 tools/diag/ab_time.py end2end_amd/csrc/libe2e_ctc.so build/diag/ab_noalign.so 2>&1 | grep -v amdgpu | tee gpurun_out/icache/ab_time.txt
for v in aligned noalign; do
  lib=$GRAFT_REPO_ROOT/end2end_amd/csrc/libe2e_ctc.so; [ $v = noalign ] && lib=$GRAFT_REPO_ROOT/build/diag/ab_noalign.so
  export E2E_CTC_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/icache/stats_$v -o k -- python3 tools/diag/time_fast_kernels.py > gpurun_out/icache/stats_$v.log 2>&1
  rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/icache/pmc_sq_$v -o pmc -- python3 tools/diag/time_fast_kernels.py > gpurun_out/icache/pmc_sq_$v.log 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d gpurun_out/icache/pmc_sqc_$v -o pmc -- python3 tools/diag/time_fast_kernels.py > gpurun_out/icache/pmc_sqc_$v.log 2>&1
  python3 tools/diag/pmc_summary.py gpurun_out/icache/pmc_sq_$v 2>&1 | tail -8
  python3 tools/diag/pmc_summary.py gpurun_out/icache/pmc_sqc_$v 2>&1 | tail -8
  find gpurun_out/icache/stats_$v -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-160
done
