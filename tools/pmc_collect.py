"""Copies the evidence of a tools/pmc_run.sh run into profiles/ (tracked): the parsed JSON and, per pass, the
rocprofv3 CSV restricted to this repo's kernels (read and written with the csv module: kernel names contain commas).

    python tools/pmc_collect.py gpurun_out/<tag> profiles/r02
"""
import csv
import glob
import os
import shutil
import sys

csv.field_size_limit(1 << 30)
KEEP = ("sa_mid_fwd", "sa_dgrad", "sa_wgrad", "mha_", "fps_", "rel_tail", "rel_fused", "relation_", "sa_l1", "sa_l3", "sa_pool", "tf_ffn", "tf_rows")


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    shutil.copy(os.path.join(src, "pmc.json"), prefix + "_pmc.json")
    for name, pat in (("fetch", "*counter_collection.csv"), ("write", "*counter_collection.csv"), ("mfma", "*counter_collection.csv"),
                      ("trace", "*kernel_trace.csv")):
        files = glob.glob(os.path.join(src, name, "**", pat), recursive=True)
        if not files:
            continue
        rows = list(csv.DictReader(open(files[0], newline="")))
        rows = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
        out = f"{prefix}_pmc_{name}.csv"
        with open(out, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
        print(out, len(rows), "rows")
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, prefix + "_pmc_kernel_stats.csv")


if __name__ == "__main__":
    main()
