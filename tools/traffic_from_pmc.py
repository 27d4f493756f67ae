"""profiles/<tag>_pmc_{fetch,write}.summary.txt -> profiles/traffic.json: HBM-side bytes per launch of the bench's
dominant kernel groups (FETCH_SIZE / WRITE_SIZE are in KB; gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE
counts the 128-byte requests of wide coalesced reads as 64 B, so it is doubled)."""
import json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
groups = {"conv_wino": "conv3x3_wino_kernel", "fc": "gemm_c8_pf_kernel", "conv_direct": "conv3x3_first_kernel", "roi_pool": "roi_pool_pm_kernel"}
def load(path, key):
    out = {}
    for line in open(path):
        m = re.search(r"n=(\d+)\s.*%s=([0-9.e+]+)" % key, line)
        if m:
            for g, pat in groups.items():
                if pat in line:
                    n, v = int(m.group(1)), float(m.group(2))
                    a = out.setdefault(g, [0, 0.0])
                    a[0] += n; a[1] += n * v
    return out
f = load("profiles/%s_pmc_fetch.summary.txt" % tag, "FETCH_SIZE")
w = load("profiles/%s_pmc_write.summary.txt" % tag, "WRITE_SIZE")
res = {}
for g in groups:
    if g in f and g in w:
        fetch = 2.0 * f[g][1] / f[g][0] * 1024
        write = w[g][1] / w[g][0] * 1024
        res[g] = {"fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write), "bytes_per_launch": round(fetch + write),
                  "launches_sampled": f[g][0]}
res["_source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline; summaries in profiles/%s_pmc_fetch.summary.txt and _write; FETCH_SIZE doubled (gfx950 128-B request correction)" % tag)
json.dump(res, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
