"""Fold rocprofv3 --pmc counter_collection CSVs into one per-kernel, per-launch table (tools/profile_round.sh)."""
import collections, csv, glob, json, os, sys


PLAN = {}          # bench.py config.launch_plan: launches of the macroblock loop's I / P / B instantiations before / inside the timed window


def mb_kind(k):
    """which instantiation of the macroblock loop a kernel name is: k_mb_slice<M, ME, PS, RD, BS>"""
    if "k_mb_slice" not in k:
        return None
    args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
    if len(args) >= 5 and args[4] in ("true", "1"):
        return "B"
    return "P" if len(args) >= 3 and args[2] in ("true", "1") else "I"


def window_of(k, v, skip, keep):
    """the launches of kernel k (sorted dispatch ids / intervals v) that lie in bench.py's timed window"""
    kind = mb_kind(k)
    if kind and PLAN.get(kind):
        w, t = PLAN[kind]["warmup"], PLAN[kind]["timed"]
        return v[w:w + t]
    return v[skip:skip + keep] if len(v) >= skip + keep else v[-keep:]


def fold(d, skip, keep):
    """per kernel: counters summed over its launches `skip` .. `skip + keep - 1` (the timed pictures of bench.py: warmup launches and the
    IDR probe at the end are left out, so the averages are over the same launches as bench.py's own event timing)"""
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not fs:
        return agg, disp
    rows = list(csv.DictReader(open(fs[0])))
    order = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        i = int(r["Dispatch_Id"])
        if i not in order[k]:
            order[k].append(i)
    # the macroblock loop has an I-slice and a P-slice instantiation: a kernel with fewer launches than warmup + timed ran only in part of the
    # pictures, and its timed launches are its last `keep` ones (the P instantiation: one warmup P picture, then the timed ones)
    window = {k: set(window_of(k, sorted(v), skip, keep)) if "k_csp" not in k else set(v) for k, v in order.items()}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Dispatch_Id"]) not in window[k]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in order:            # diagnostics for the round's log: which dispatches of the macroblock loop were folded
        if "k_mb_slice" in k:
            per = collections.defaultdict(dict)
            for r in rows:
                if r["Kernel_Name"].split("(")[0].replace("void ", "") == k:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            print("fold", os.path.basename(d), k, "dispatches", sorted(per), "window", sorted(window[k]), "grid", sorted({r.get("Grid_Size", "?") for r in rows if r["Kernel_Name"].split("(")[0].replace("void ", "") == k}),
                  {i: {c: "%.3g" % x for c, x in list(v.items())[:2]} for i, v in sorted(per.items())})
    return agg, disp


def trace_window(d, skip, keep):
    """average duration per kernel over the same window of launches, from the --kernel-trace CSV"""
    fs = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    if not fs:
        return {}
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = {}
    for k, v in per.items():
        v.sort()
        w = window_of(k, v, skip, keep) if "x264gpu" in k and "k_csp" not in k else v
        if w:
            out[k] = {"launches": len(w), "avg_ms": sum(e - s for s, e in w) / len(w) / 1e6, "all_launches": len(v), "avg_ms_all_launches": sum(e - s for s, e in v) / len(v) / 1e6}
    return out


def main():
    out = sys.argv[1]
    table = collections.defaultdict(dict)
    try:
        bj = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().split("\n")[-1])
        skip, keep = bj["warmup"], bj["steps"]
        PLAN.update(bj["config"].get("launch_plan", {}))
    except Exception:  # noqa: BLE001
        skip, keep = 0, 1 << 30
    kt = trace_window(sys.argv[2], skip, keep)
    json.dump(kt, open(os.path.join(out, "kernel_trace_timed_window.json"), "w"), indent=1, sort_keys=True)
    for d in sys.argv[3:]:
        agg, disp = fold(d, skip, keep)
        for k, v in agg.items():
            n = max(len(disp[k]), 1)
            table[k].setdefault("launches_profiled", {})
            for c, x in v.items():
                table[k][c] = x / n
                table[k]["launches_profiled"][c] = n
    # the macroblock loop as a whole: its instantiations' counters summed over the timed window, per launch of that window
    mbk = [k for k in table if mb_kind(k)]
    if mbk:
        tot, n = collections.defaultdict(float), 0
        for k in mbk:
            nk = max(table[k]["launches_profiled"].values())
            n += nk
            for c, x in table[k].items():
                if c != "launches_profiled":
                    tot[c] += x * table[k]["launches_profiled"][c]
        table["_mb_loop_timed_window"] = {c: x / n for c, x in tot.items()}
        table["_mb_loop_timed_window"]["launches_profiled"] = {c: n for c in tot}
        table["_mb_loop_timed_window"]["instantiations"] = {k: max(table[k]["launches_profiled"].values()) for k in mbk}
        kts = {k: kt[k] for k in kt if mb_kind(k)}
        if kts:
            table["_mb_loop_timed_window"]["avg_ms_kernel_trace"] = sum(v["avg_ms"] * v["launches"] for v in kts.values()) / sum(v["launches"] for v in kts.values())
    for k, v in table.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            # rocprofv3 reports KB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads
            # (MI355X_MICROARCH.md "HBM"): the read side is doubled, which is an upper bound for narrower accesses.
            v["hbm_bytes_per_launch_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            v["hbm_bytes_per_launch_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        # (SQ_WAVES itself reads 8/7 of the launched wavefronts on this part — 2340.57 for 2048 one-wave workgroups, on every dispatch —
        #  while the instruction and cycle counters add up for 2048: per-wave figures are not derived from it)
    # streams (= frames) one launch covers in the profiled command, so bench.py only quotes these
    # figures for the same workload
    try:
        b = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().split("\n")[-1])
        table["_workload"] = {"streams_per_launch": b["config"]["streams_per_gpu"], "config": b["config"]["workload"], "toolset": b["config"]["toolset"], "content": b["config"].get("content", "noise")}
        for k, v in table.items():
            if not k.startswith("_"):
                v["macroblocks_per_launch"] = b["config"]["streams_per_gpu"] * b["config"]["mb_per_frame"]
    except Exception as e:  # noqa: BLE001
        table["_workload"] = {"streams_per_launch": None, "error": str(e)}
    json.dump(table, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(table.items()):
        if k.startswith("_"):
            continue
        print(k, {c: ("%.4g" % x if isinstance(x, float) else x) for c, x in v.items() if c != "launches_profiled"})


if __name__ == "__main__":
    main()
