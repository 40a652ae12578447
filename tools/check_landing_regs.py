"""Build-time check for the streaming split-bf16 kernels (spacap3d_amd/csrc/sa_bf3.inc): their prefetched rows land in
AGPRs a0..a63 through inline asm, invisibly to the compiler.  That is only sound if no compiler-generated instruction of
those kernels touches a0..a63.  This script compiles sa_mlp.hip to assembly and fails if one does.

    python tools/check_landing_regs.py        (needs hipcc; no GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spacap3d_amd", "csrc")
LANDING = 64
KERNELS = ("sa_mid_fwd_bf3s_kernel", "sa_mid_fwd_s_kernel", "sa_dgrad_s_kernel", "sa_wgrad_s_kernel")


def check(asm_text):
    bad, seen = [], 0
    for m in re.finditer(r"^(_ZN\S+):[^\n]*\n(.*?)^\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not any(k in name for k in KERNELS):
            continue
        seen += 1
        inasm = False
        for n, line in enumerate(body.split("\n")):
            if "ASMSTART" in line:
                inasm = True
            elif "ASMEND" in line:
                inasm = False
            elif not inasm:
                code = line.split(";")[0]
                for r in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
                    lo = int(r.group(1) or r.group(3))
                    if lo < LANDING:
                        bad.append((name, n, line.strip()))
    return seen, bad


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-ffp-contract=fast", "-save-temps", "-c", os.path.join(CSRC, "sa_mlp.hip"), "-o", os.path.join(tmp, "sa.o"),
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
        subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(os.path.join(tmp, "sa_mlp-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    seen, bad = check(text)
    print(f"{seen} streaming kernels checked, {len(bad)} compiler-generated uses of a0..a{LANDING - 1}")
    for name, n, line in bad[:20]:
        print("  ", name[:60], n, line)
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main())
