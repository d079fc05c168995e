"""profiles/traffic.json from a condensed PMC file (scripts/pmc_summary.py): per kernel the HBM bytes per launch that
bench.py quotes in roofline.traffic (it cannot collect counters itself).
Usage: python scripts/make_traffic_json.py profiles/round2_v6_bench_pmc.csv"""
import csv, json, os, re, sys

src = sys.argv[1]
out = {}
for r in csv.DictReader(open(src)):
    m = re.search(r"(?:\)::|^|\s)(\w+?)(?:<[^(]*)?\(", r["kernel"])
    name = m.group(1) if m else r["kernel"][:40]
    if name in out or not r.get("hbm_bytes_per_dispatch"):
        continue
    out[name] = {
        "hbm_bytes_per_launch": int(float(r["hbm_bytes_per_dispatch"])),
        "fetch_size_kib": float(r["FETCH_SIZE"]),
        "write_size_kib": float(r["WRITE_SIZE"]),
        "mfma_util_pct": float(r["MfmaUtil_pct"] or 0),
        "valu_per_mfma": (round(float(r["SQ_INSTS_VALU"]) / float(r["SQ_INSTS_MFMA"]), 2)
                          if r.get("SQ_INSTS_VALU") and r.get("SQ_INSTS_MFMA") and float(r["SQ_INSTS_MFMA"]) > 0 else None),
        "avg_us_profiled": float(r["avg_us_profiled"]) if r.get("avg_us_profiled") else None,
        "source": f"{src} (scripts/gpu_profile_bench.sh: rocprofv3 --pmc passes over `bench.py --lean --match-streams 1 --cfg5-sample 2000 --steps 4 --warmup 1`, "
                  "FETCH_SIZE and WRITE_SIZE in passes of their own; hbm = (2*FETCH+WRITE)*1024, MI355X_MICROARCH.md HBM section)",
    }
out["_source"] = os.path.basename(src)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(list(out.keys()))
