"""Launches every case of tools/kernel_cases.py a few times -- the workload for rocprofv3 passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/pmc_kernels.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/pmc_kernels.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
              --output-format csv -d $OUT/mfma -- python3 $REPO/tools/pmc_kernels.py
    python tools/pmc_parse.py $OUT profiles/r02_pmc.json          # -> the file bench.py reads `traffic` from

(one counter group per pass: FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together, MI355X_MICROARCH.md
"rocprofv3 PMC slots"; no other tracing in a --pmc pass).  A marker file lists the cases in launch order with the number
of launches each, so the parser can attribute dispatches to cases by kernel name AND order.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_cases as KC  # noqa: E402

REPS = 5


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    order = []
    for make in KC.cases(dev):
        case = make()
        for _ in range(REPS):
            case["run"]()
        torch.cuda.synchronize()
        order.append(dict(name=case["name"], kernel=case["kernel"], launches=REPS, flops=case["flops"], bytes=case["bytes"]))
        del case
        torch.cuda.empty_cache()
    out = os.environ.get("PMC_ORDER_FILE")
    if out:
        json.dump(order, open(out, "w"), indent=1)
    print(json.dumps(order))


if __name__ == "__main__":
    main()
