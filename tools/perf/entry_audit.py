"""What every kernel of libe2e_ctc.so does before its first barrier / branch: scratch traffic at the entry.

Why: a kernel whose parameter block has its address taken -- handed by reference to a function the compiler then decides NOT to
inline -- keeps the block in private memory: every wave copies the whole kernarg segment (ExactParams: 336 bytes per LANE, 21 KB per
wave) to scratch at its entry, before any early exit, and reads its fields back from there.  That is what made the flagged-utterance
launch take 11.0 instead of 5.2 us in some builds of round 5 (profiles/r05_placement/) and, by every sign, what round 4 met as an
"unchanged kernel 6.5x slower after a neighbour grew" (inlining decisions move with code size).  The check needs no GPU: it
disassembles the device code of the built library.

  python tools/perf/entry_audit.py [path/to/libe2e_ctc.so]      -> one line per kernel: scratch stores / loads in its first 200 instructions
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(MAGIC, data)]
    out = []
    for i, o in enumerate(offs):
        end = offs[i + 1] if i + 1 < len(offs) else len(data)
        b = os.path.join(tmp, "b%d.bin" % i); c = os.path.join(tmp, "co%d.o" % i)
        open(b, "wb").write(data[o:end])
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + b,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + c], capture_output=True)
        if r.returncode == 0 and os.path.getsize(c) > 0: out.append(c)
    return out


def audit(lib, window=200):
    """{kernel name (demangled): (scratch stores, scratch loads) among its first `window` instructions}"""
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True).stdout
            kernels = {l.split()[-1][:-3] for l in syms.splitlines() if l.strip().endswith(".kd")}
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur, n = None, 0
            for l in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", l)
                if m:
                    cur = m.group(1) if m.group(1) in kernels else None; n = 0
                    if cur: res[cur] = [0, 0]
                    continue
                if cur is None or not l.startswith("\t"): continue
                n += 1
                if n > window: cur = None; continue
                op = l.split()[0]
                if op.startswith("scratch_store"): res[cur][0] += int(re.search(r"x(\d)$", op).group(1)) if re.search(r"x(\d)$", op) else 1
                elif op.startswith("scratch_load"): res[cur][1] += 1
    names = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"e2e::\(anonymous namespace\)::|e2e::fastk::\(anonymous namespace\)::|e2e::fastk::|e2e::", "", nm): tuple(v)
            for nm, v in zip(names, res.values())}


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "end2end_amd", "csrc", "libe2e_ctc.so")
    r = audit(lib)
    for k in sorted(r, key=lambda k: -r[k][0]):
        print("%4d dwords stored to scratch, %3d scratch loads   %s" % (r[k][0], r[k][1], k[:150]))
