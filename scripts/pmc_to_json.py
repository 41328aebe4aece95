"""profiles/<tag>_pmc_traffic.txt (FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_final.sh) -> profiles/pmc_traffic.json,
the per-launch HBM traffic bench.py quotes in its roofline object.  Correction per MI355X_MICROARCH.md "HBM": on
gfx950 FETCH_SIZE counts 128-byte requests at 64 B for wide (16 B/lane) coalesced streaming reads -> doubled for the
kernels whose reads are of that kind (the pre-filter GEMM, the feature-row kernels); narrow-access kernels keep the raw
value (uncalibrated, flagged)."""
import json, re, sys
WIDE = ("k_filter", "tk_mid2", "tk_apply23", "tk_front", "tk_mid1")
src = sys.argv[1]
out = {}
wl = ctr = None
for line in open(src):
    m = re.match(r"# (\w+)\s+(FETCH_SIZE|WRITE_SIZE)", line)
    if m:
        wl, ctr = m.group(1), m.group(2)
        continue
    m = re.match(r"(.+?)\s+([0-9.]+)\s+\((\d+) launches\)", line)
    if m and wl:
        name = m.group(1).strip()
        key = name.split("(")[0].replace("void ", "").strip()
        d = out.setdefault(wl, {}).setdefault(key, {})
        d["fetch_kib_raw" if ctr == "FETCH_SIZE" else "write_kib_raw"] = float(m.group(2))
for wl, ks in out.items():
    for k, d in ks.items():
        wide = any(k.startswith(w) for w in WIDE)
        f = d.get("fetch_kib_raw", 0.0) * (2.0 if wide else 1.0)
        d["fetch_corrected_x2"] = wide
        d["traffic_bytes"] = (f + d.get("write_kib_raw", 0.0)) * 1024.0
json.dump({"source": src, "unit": "bytes per launch", "workloads": out}, open("profiles/pmc_traffic.json", "w"), indent=1)
print("wrote profiles/pmc_traffic.json", {w: len(k) for w, k in out.items()})
