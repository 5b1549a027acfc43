"""Histogram of rocprofv3 PC samples (tools/pc_sample.sh): by kernel, by source line (the line-table comment rocprofv3 attaches to a sampled
instruction), by instruction class.  Writes hist.json + the first raw lines (so that a reader can check the columns) into the output directory."""
import csv
import glob
import json
import os
import re
import sys
from collections import Counter, defaultdict


def main():
    raw, out = sys.argv[1], sys.argv[2]
    files = sorted(glob.glob(os.path.join(raw, "**", "*pc_sampling*.csv"), recursive=True))
    ktrace = sorted(glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True))
    disp = {}
    for f in ktrace:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Dispatch_Id") or row.get("Correlation_Id")
                disp[k] = re.sub(r"\(.*", "", row.get("Kernel_Name", "?"))[:120]
    res = {"files": files, "n": 0}
    by_kernel, by_line, by_inst = Counter(), defaultdict(Counter), defaultdict(Counter)
    head = []
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            res["columns"] = cols
            c_inst = next((c for c in cols if c.lower() == "instruction"), None)
            c_cmt = next((c for c in cols if "comment" in c.lower()), None)
            c_disp = next((c for c in cols if c.lower() == "dispatch_id"), None)
            for i, row in enumerate(rd):
                if len(head) < 40:
                    head.append(row)
                kn = disp.get(row.get(c_disp, ""), "?") if c_disp else "?"
                by_kernel[kn] += 1
                line = (row.get(c_cmt) or "?") if c_cmt else "?"
                line = re.sub(r"^.*/x264vfw_amd/", "", line)
                by_line[kn][line] += 1
                inst = (row.get(c_inst) or "?").split(" ")[0] if c_inst else "?"
                by_inst[kn][inst] += 1
                res["n"] += 1
    res["by_kernel"] = by_kernel.most_common(40)
    res["by_line"] = {k: v.most_common(400) for k, v in by_line.items() if by_kernel[k] * 50 >= max(res["n"], 1)}
    res["by_inst"] = {k: v.most_common(60) for k, v in by_inst.items() if by_kernel[k] * 50 >= max(res["n"], 1)}
    with open(os.path.join(out, "hist.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    with open(os.path.join(out, "head.json"), "w") as fh:
        json.dump(head, fh, indent=1)
    print("samples", res["n"], "kernels", by_kernel.most_common(6))


if __name__ == "__main__":
    main()
