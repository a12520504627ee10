"""profiles/traffic_latest.json from a summary of tools/collect_profiles.sh (run here, after gpurun merged it back):
  python tools/update_traffic.py gpurun_out/prof_r02/summary_r02.json profiles/r02/summary_r02.json
HBM bytes per launch = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (separate --pmc passes; the x2 is the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md "HBM").  Keyed by the kernel's base symbol (the instantiation that moves the most bytes) and by
the full instantiation name; bench.py copies the figure of its dominant kernel into roofline.traffic and names this commit."""
import json, os, re, subprocess, sys

src, kept = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summ = json.load(open(src))
by_full = {k: v["hbm_bytes_corrected"] for k, v in summ["traffic_bytes_per_launch"].items()}
by_sym = {}
for k, b in by_full.items():
    base = re.sub(r"<.*", "", k.replace("br::", "").replace("(anonymous namespace)::", ""))
    by_sym[base] = max(b, by_sym.get(base, 0.0))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "binary-recommendation_amd", "bench.py"], cwd=root, capture_output=True, text=True).stdout.strip())
out = {"hbm_bytes_per_launch": by_sym, "by_instantiation": by_full, "commit": commit + ("+uncommitted changes" if dirty else ""),
       "source": f"{kept}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                 "--no-lazy --no-legs`; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction, MI355X_MICROARCH.md HBM)"}
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
os.makedirs(os.path.dirname(os.path.join(root, kept)), exist_ok=True)
json.dump(summ, open(os.path.join(root, kept), "w"), indent=1)
print(json.dumps(by_sym, indent=1))
