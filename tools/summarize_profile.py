"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into one JSON summary:
per-kernel time (from --stats), HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, in bytes — the
gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE reads exactly half the bytes of a wide
coalesced stream; WRITE_SIZE is exact for 16-B-per-lane stores) and SQ counters of the MFMA kernels."""
import csv, glob, json, os, sys, collections

out_dir, tag = sys.argv[1], sys.argv[2]

def first(pattern):
    g = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return g[0] if g else None

def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:80]

summary = {"tag": tag, "kernels": [], "traffic_bytes_per_launch": {}, "sq": {}}
st = first("trace/**/*kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st)):
        summary["kernels"].append({"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                   "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])})
def counter_avg(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return acc
f, w = first("fetch/**/*counter_collection.csv"), first("write/**/*counter_collection.csv")
if f and w:
    fa, wa = counter_avg(f), counter_avg(w)
    for k in fa:
        if "br::" not in k and "lookup_sort" not in k and "chunk_" not in k: continue
        # the sweep kernel runs on two table sizes: keep the largest-grid launches separately
        fv, wv = fa[k].get("FETCH_SIZE", []), wa.get(k, {}).get("WRITE_SIZE", [])
        if not fv: continue
        # kernels launched on two table sizes with the same (capped) grid: keep the larger table's
        # launches = the upper half by counter value
        def upper(vals):
            xs = sorted(v for v, _ in vals)
            if len(xs) >= 2 and xs[-1] > 1.5 * xs[0]:
                xs = xs[len(xs) // 2:]
            return xs
        fsel, wsel = upper(fv), upper(wv)
        fetch_kib = sum(fsel) / len(fsel); write_kib = (sum(wsel) / len(wsel)) if wsel else 0.0
        summary["traffic_bytes_per_launch"][k] = {"FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                                                  "hbm_bytes_corrected": 2 * fetch_kib * 1024 + write_kib * 1024, "launches": len(fsel)}
s = first("sq/**/*counter_collection.csv")
if s:
    sa = counter_avg(s)
    for k, d in sa.items():
        if "dense_" in k or "inbatch" in k:
            summary["sq"][k] = {c: sum(v for v, _ in vals) / len(vals) for c, vals in d.items()}
print(json.dumps(summary, indent=1))
