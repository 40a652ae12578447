"""Lab: where does sa_mid_fwd spend its time?  SPACAP_SA_LAB = 0 (normal) / 1 (no MFMA) / 2 (no global stores) / 3 (no
global loads after the first tile), one process per variant (the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, kernel_cases as KC
    dev = torch.device("cuda:0")
    for R, ci, co in ((262144, 128, 256), (262144, 128, 128), (1048576, 64, 64), (1048576, 64, 128)):
        c = KC.sa_mid_fwd(R, ci, co, dev, "lab")
        us = KC.time_case(c)
        print(f"LAB={os.environ.get('SPACAP_SA_LAB', '0')} {ci:3d}->{co:3d} R={R:8d}: {us:7.1f} us  {c['flops'] / us * 1e-6:6.1f} TF/s  {c['bytes'] / us * 1e-3:7.1f} GB/s", flush=True)
        del c
else:
    for lab in (sys.argv[2] if len(sys.argv) > 2 else "01234"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, SPACAP_SA_LAB=lab))
