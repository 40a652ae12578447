"""Build-time check for the streaming shared-MLP kernels (spacap3d_amd/csrc/sa_bf3.inc, sa_stream.inc).

Their prefetched rows land in AGPRs through inline-asm loads; the C++ code sees them only after a hand-counted s_waitcnt whose
"+a" operands name the same registers.  The register allocator knows the values are live in between, but nothing stops it from
COPYING a landing register (or spilling into one it considers dead on some path) while the data is still on its way.  This
script compiles sa_mlp.hip to assembly and checks, per kernel, in layout order:

  * no compiler-generated instruction ever WRITES a landing register,
  * no compiler-generated instruction READS one between a load block that targets it and the wait block that names it
    (state carried round the loop: a register counts as in flight from the top of the function until its first wait).

    python tools/check_landing_regs.py                  compiles sa_mlp.hip itself (needs hipcc; no GPU)
    python tools/check_landing_regs.py --asm FILE.s     checks the assembly the build kept (csrc/Makefile runs this right
                                                        after compiling sa_mlp.o and deletes the object when it fails)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spacap3d_amd", "csrc")
KERNELS = ("sa_mid_fwd_bf3s_kernel", "sa_mid_fwd_s_kernel", "sa_dgrad_bf3s_kernel")
AREG = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def aregs(text):
    out = set()
    for m in AREG.finditer(text):
        lo = int(m.group(1) or m.group(3))
        hi = int(m.group(2) or m.group(3))
        out.update(range(lo, hi + 1))
    return out


def check_kernel(name, body):
    lines = body.split("\n")
    # pass 1: the landing registers = destinations of loads inside asm blocks
    landing, inasm = set(), False
    for line in lines:
        if "ASMSTART" in line:
            inasm = True
        elif "ASMEND" in line:
            inasm = False
        elif inasm and "global_load" in line:
            landing |= aregs(line.split(",")[0])
    if not landing:
        return [(name, 0, "no landing registers found")]
    bad, inflight, inasm = [], set(landing), False
    for n, line in enumerate(lines):
        if "ASMSTART" in line:
            inasm = True
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0]
        if inasm:
            if "global_load" in code:
                inflight |= aregs(code.split(",")[0])
            elif "s_waitcnt" in code and "landed" in line:
                inflight -= aregs(line.split("landed")[1])
            continue
        regs = aregs(code) & landing
        if not regs:
            continue
        ops = code.split(None, 1)
        dst = aregs(ops[1].split(",")[0]) if len(ops) > 1 else set()
        if dst & landing:
            bad.append((name, n, "writes a landing register: " + line.strip()))
        elif regs & inflight:
            bad.append((name, n, "reads a landing register in flight: " + line.strip()))
    return bad


def check(asm_text):
    seen, bad = 0, []
    for m in re.finditer(r"^(_ZN\S+):[^\n]*\n(.*?)^\.Lfunc_end", asm_text, re.S | re.M):
        if any(k in m.group(1) for k in KERNELS):
            seen += 1
            bad += check_kernel(m.group(1), m.group(2))
    return seen, bad


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--asm":
        seen, bad = check(open(sys.argv[2]).read())
        ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0]
        print(f"{seen} streaming kernels checked, {len(bad)} unsafe uses of landing registers  [{ver.strip()}]")
        for name, n, what in bad[:20]:
            print("  ", name[:60], n, what)
        return 1 if bad or not seen else 0
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-ffp-contract=fast", "-save-temps", "-c", os.path.join(CSRC, "sa_mlp.hip"), "-o", os.path.join(tmp, "sa.o"),
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
        subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(os.path.join(tmp, "sa_mlp-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    seen, bad = check(text)
    print(f"{seen} streaming kernels checked, {len(bad)} unsafe uses of landing registers")
    for name, n, what in bad[:20]:
        print("  ", name[:60], n, what)
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main())
