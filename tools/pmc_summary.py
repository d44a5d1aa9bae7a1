"""Per-kernel means of rocprofv3 --pmc counters (one counter_collection.csv per pass directory).

usage: python tools/pmc_summary.py OUT.json DIR [DIR...]
Each DIR is searched for *_counter_collection.csv.  Counter values are averaged per launch and per kernel (short name).
HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads, so the read side is doubled
("fetch_bytes_corrected"); WRITE_SIZE is taken as reported.
MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (32 per
v_mfma_f32_32x32x16, 16 per 16x16x32); GRBM_GUI_ACTIVE is the elapsed GPU clock per counter instance.  mfma_util =
MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD; the CSV row holds the sum over the 8 XCDs).  Profiled runs clock lower than un-profiled ones.
"""
import collections, csv, glob, json, os, re, sys

csv.field_size_limit(1 << 30)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:100]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = collections.defaultdict(float)  # (dispatch, kernel, counter) -> summed over XCD/SE instances
            inst = collections.defaultdict(int)
            dur = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                inst[key] += 1
                dur[(r["Dispatch_Id"], short(r["Kernel_Name"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            for (d_, k, c), v in per_dispatch.items():
                a = agg[k][c]
                a[0] += 1
                a[1] += v
                if c == "GRBM_GUI_ACTIVE":  # rocprofv3 reports ONE row per dispatch: the sum over the 8 XCDs (19.7 "GHz" otherwise)
                    b = agg[k]["GRBM_GUI_ACTIVE_per_instance"]
                    b[0] += 1
                    b[1] += v / (inst[(d_, k, c)] if inst[(d_, k, c)] > 1 else 8)
            for (d_, k), us in dur.items():
                a = agg[k]["profiled_us"]
                a[0] += 1
                a[1] += us
    res = {}
    for k, cs in agg.items():
        e = {c: v[1] / v[0] for c, v in cs.items()}
        e["launches"] = max(v[0] for c, v in cs.items() if c not in ("profiled_us", "GRBM_GUI_ACTIVE_per_instance"))
        if "FETCH_SIZE" in e:
            e["fetch_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE_per_instance", 0) > 0:
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE_per_instance"])
        res[k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0))[:25]:
        print(f"{k[:70]:70s} n={e['launches']:5d} " + " ".join(f"{c}={v:.4g}" for c, v in e.items() if c != "launches"))


if __name__ == "__main__":
    main()
