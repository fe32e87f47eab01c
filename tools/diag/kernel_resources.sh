#!/bin/bash
# registers / scratch / spills of every kernel of one source, as the compiler reports them:
#   tools/diag/kernel_resources.sh ctc_loss_fast.hip [grep-pattern] ["-DFOO"]
cd "$(dirname "$0")/../../end2end_amd/csrc"
extra=; [[ $1 == ctc_loss_fast*.hip ]] && extra=-fno-slp-vectorize
/opt/rocm/bin/hipcc $extra -O3 -std=c++17 -fPIC --offload-arch=gfx950 $3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c $1 -o /dev/null 2>&1 |
  python3 -c '
import sys, re, subprocess
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", ln) or re.search(r"(Function Name|Name): (\S+)", ln)
    if m: cur = m.group(2); rows[cur] = {}; continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur: rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"e2e::\(anonymous namespace\)::|e2e::fastk::|e2e::", "", name)
    if pat in name:
        print("%-86s vgpr %3d agpr %3d scratch %4d sspill %3d vspill %3d occ %d lds %d" % (name[:86], r.get("VGPRs", -1), r.get("AGPRs", 0),
              r.get("ScratchSize", 0), r.get("SGPRs Spill", 0), r.get("VGPRs Spill", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
' "$2"
