"""Fold rocprofv3 --pmc counter_collection CSVs into one per-kernel, per-launch table (tools/profile_round.sh)."""
import collections, csv, glob, json, os, sys


def fold(d):
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    if not fs:
        return agg, disp
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, disp


def main():
    out = sys.argv[1]
    table = collections.defaultdict(dict)
    for d in sys.argv[2:]:
        agg, disp = fold(d)
        for k, v in agg.items():
            n = max(len(disp[k]), 1)
            table[k].setdefault("launches_profiled", {})
            for c, x in v.items():
                table[k][c] = x / n
                table[k]["launches_profiled"][c] = n
    for k, v in table.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            # rocprofv3 reports KB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads
            # (MI355X_MICROARCH.md "HBM"): the read side is doubled, which is an upper bound for narrower accesses.
            v["hbm_bytes_per_launch_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            v["hbm_bytes_per_launch_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v and v["SQ_WAVES"]:
            v["valu_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
            v["cycles_per_wave"] = v["SQ_WAVE_CYCLES"] / v["SQ_WAVES"]
    # frames one launch covers in the profiled command (bench.py default: streams / groups), so bench.py only quotes these
    # figures for the same workload
    try:
        b = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().split("\n")[-1])
        table["_workload"] = {"frames_per_launch": b["config"]["streams_per_gpu"] // b["config"]["stream_groups"], "config": b["config"]["workload"]}
    except Exception as e:  # noqa: BLE001
        table["_workload"] = {"frames_per_launch": None, "error": str(e)}
    json.dump(table, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(table.items()):
        if k.startswith("_"):
            continue
        print(k, {c: ("%.4g" % x if isinstance(x, float) else x) for c, x in v.items() if c != "launches_profiled"})


if __name__ == "__main__":
    main()
