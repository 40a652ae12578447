"""Turns the rocprofv3 CSV output of the tools/pmc_kernels.py passes into one small tracked JSON.

    python tools/pmc_parse.py <dir with trace/ fetch/ write/ mfma/ sub-directories> <out.json>

Kernel names contain commas (template arguments), so the files are read with the csv module, never split on ','.
Per case (tools/kernel_cases.py), the LAST launch of the case's kernel sequence is used (warm caches, as in the step):
  fetch_bytes_corrected = 2 * FETCH_SIZE * 1024   (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM)
  write_bytes           = WRITE_SIZE * 1024
  hbm_bytes             = the sum of the two = `traffic` of the bench line
  mfma_util             = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): share of the chip's SIMD cycles
                          with an MFMA executing (BUSY_CYCLES counts per-SIMD cycles: exactly flops / 64 for
                          v_mfma_f32_16x16x4_f32; GRBM_GUI_ACTIVE is summed over the 8 XCDs) -- equals achieved / peak
                          TFLOP/s at the clock the kernel actually ran at (GRBM_GUI_ACTIVE / 8 / duration, ~2.26 GHz
                          under these kernels vs the 2.4 GHz of the 157.3 TFLOP/s spec figure)
  mfma_flops_counted    = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512   (matches the algorithmic flops of every GEMM case)
"""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def rows_of(d, suffix):
    files = glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True)
    out = []
    for f in files:
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def per_dispatch_counters(d):
    """{dispatch_id: {"kernel": name, counter: value}} from a counter_collection.csv (one row per dispatch x counter)."""
    disp = collections.OrderedDict()
    for r in rows_of(d, "counter_collection.csv"):
        did = int(r["Dispatch_Id"])
        e = disp.setdefault(did, {"kernel": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    order = json.load(open(os.path.join(src, "order.json")))
    result = {"_how": "tools/pmc_kernels.py under rocprofv3, parsed by tools/pmc_parse.py; see the docstrings for the formulas",
              "_cases": {}}
    passes = {}
    for name in ("fetch", "write", "mfma"):
        p = os.path.join(src, name)
        if os.path.isdir(p):
            passes[name] = per_dispatch_counters(p)
    trace = sorted(rows_of(os.path.join(src, "trace"), "kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))

    def pick(seq, key):
        """dispatches of the hot kernels in launch order -> per case the last launch."""
        hot = [d for d in seq if any(o["kernel"].split("<")[0] in d[key] for o in order)]
        out, i = {}, 0
        for o in order:
            k = o["kernel"].split("<")[0]
            mine = []
            while i < len(hot) and len(mine) < o["launches"]:
                if k in hot[i][key]:
                    mine.append(hot[i])
                i += 1
            out[o["name"]] = mine
        return out

    tr = pick(trace, "Kernel_Name") if trace else {}
    pk = {n: pick(seq, "kernel") for n, seq in passes.items()}
    for o in order:
        rec = {"flops": o["flops"], "algorithmic_bytes": o["bytes"]}
        if tr.get(o["name"]):
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr[o["name"]]]
            rec["profiled_us"] = sum(durs[1:]) / max(1, len(durs) - 1)
            rec["kernel_name"] = tr[o["name"]][-1]["Kernel_Name"][:160]
        f = pk.get("fetch", {}).get(o["name"])
        w = pk.get("write", {}).get(o["name"])
        m = pk.get("mfma", {}).get(o["name"])
        if f:
            rec["FETCH_SIZE_KB"] = f[-1].get("FETCH_SIZE")
            rec["fetch_bytes_corrected"] = 2.0 * f[-1].get("FETCH_SIZE", 0.0) * 1024
        if w:
            rec["WRITE_SIZE_KB"] = w[-1].get("WRITE_SIZE")
            rec["write_bytes"] = w[-1].get("WRITE_SIZE", 0.0) * 1024
        if f and w:
            rec["hbm_bytes"] = rec["fetch_bytes_corrected"] + rec["write_bytes"]
        if m:
            c = m[-1]
            for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                      "GRBM_GUI_ACTIVE"):
                if k in c:
                    rec[k] = c[k]
            if c.get("GRBM_GUI_ACTIVE"):
                rec["mfma_util"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                # (GRBM_GUI_ACTIVE also counts the dispatch's ramp: for launches under ~50 us the quotient exceeds the chip's
                # clock -- 3.3 "GHz" was reported for a 14 us launch in round 3 -- so it is only given for long ones)
                if rec.get("profiled_us") and rec["profiled_us"] >= 50.0:
                    rec["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / rec["profiled_us"] * 1e-3
            if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
                rec["mfma_flops_counted"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
            if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
                rec["mfma_bf16_flops_counted"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
        rec["source"] = os.path.basename(dst)
        result["_cases"][o["name"]] = rec
    result.update(result.pop("_cases"))
    json.dump(result, open(dst, "w"), indent=1)
    for k, v in result.items():
        if isinstance(v, dict):
            print(f"{k:60s} us {v.get('profiled_us', 0):8.1f}  hbm {v.get('hbm_bytes', 0) / 1e6:8.1f} MB  alg {v['algorithmic_bytes'] / 1e6:8.1f} MB  "
                  f"mfma_util {v.get('mfma_util', 0):.3f}")


if __name__ == "__main__":
    main()
