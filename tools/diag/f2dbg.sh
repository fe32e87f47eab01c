for d in ${DBGS:-0 1 2 3 4 5}; do echo "dbg=$d"; E2E_F2_DBG=$d python3 tools/diag/time_fast_kernels.py 2>&1 | grep "S<=200" | head -1; done
