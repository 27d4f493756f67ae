"""profiles/<tag>_pmc_{fetch,write}.summary.txt -> profiles/traffic.json: HBM-side bytes per launch of the bench's
dominant kernel groups (FETCH_SIZE / WRITE_SIZE are in KB; gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE
counts the 128-byte requests of wide coalesced reads as 64 B, so it is doubled)."""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# --commit <hash>: the commit whose binary the PMC passes were collected on (VERDICT r5 task 5: roofline.traffic_source names it)
commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else None
COMMIT_NOTE = ("; collected on the binary of commit %s" % commit) if commit else ""
if "--config" in sys.argv:  # the other BASELINE configs: the DOMINANT kernel (by total duration) of `bench.py --config <key>` under the two PMC passes
    key = sys.argv[sys.argv.index("--config") + 1]
    images_arg = sys.argv[sys.argv.index("--images") + 1] if "--images" in sys.argv else "auto"
    def rows(path, ctr):
        out = {}
        for line in open(path):
            m = re.match(r"(.*?)\s+grid=\S+\s+n=(\d+)\s+(?:dur=\s*([0-9.]+)us)?.*?%s=([0-9.e+]+)" % ctr, line)
            if m:
                name = re.sub(r"<.*", "", m.group(1).strip().split("::")[-1])
                a = out.setdefault(name, [0, 0.0, 0.0])
                n = int(m.group(2)); a[0] += n; a[1] += n * float(m.group(3) or 0); a[2] += n * float(m.group(4)) * 1024
        return out
    f = rows("profiles/%s_%s_pmc_fetch.summary.txt" % (tag, key), "FETCH_SIZE")
    w = rows("profiles/%s_%s_pmc_write.summary.txt" % (tag, key), "WRITE_SIZE")
    images = int(images_arg) if images_arg != "auto" else sum(v[0] for k, v in f.items() if "image_transform" in k)  # one transform launch per image
    dom = max(f, key=lambda k: f[k][1])
    fb, wb = 2.0 * f[dom][2], w.get(dom, [0, 0, 0.0])[2]
    tot = sum(2.0 * v[2] for v in f.values()) + sum(v[2] for v in w.values())
    tj = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
    tj[key] = {"dominant": {"kernel": dom, "launches_per_image": f[dom][0] / images, "bytes_per_launch": round((fb + wb) / f[dom][0]),
                            "bytes_per_image": round((fb + wb) / images), "fetch_bytes_per_image": round(fb / images), "write_bytes_per_image": round(wb / images)},
               "all_kernels_bytes_per_image": round(tot / images), "images_sampled": images,
               "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --config ... --steps 2 --warmup 1 "
                          "--no-cpu-baseline --sustained-seconds 0; summaries in profiles/%s_%s_pmc_fetch.summary.txt and _write; FETCH_SIZE doubled (gfx950 128-B request correction)" % (tag, key) + COMMIT_NOTE}
    json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(tj[key], indent=1))
    sys.exit(0)
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
                  "--no-cpu-baseline; summaries in profiles/%s_pmc_fetch.summary.txt and _write; FETCH_SIZE doubled (gfx950 128-B request correction)" % tag) + COMMIT_NOTE
old = json.load(open("profiles/traffic.json")) if os.path.exists("profiles/traffic.json") else {}
for k_, v_ in old.items():   # keep the per-config entries (c1 / c3 / c4 / c5) written by --config
    if isinstance(v_, dict) and "dominant" in v_:
        res[k_] = v_
json.dump(res, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
