"""summarise a rocprofv3 --pmc --kernel-trace run: per kernel name avg duration, effective clock, MFMA util, wait fractions"""
import csv, sys, collections, glob, os
d = sys.argv[1]
ct = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
kt = glob.glob(os.path.join(d, "*kernel_trace.csv"))
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(ct)):
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); cnt[key] += 1
        acc[key]["_dur_us"] += dur.get(r["Dispatch_Id"], 0.0)
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("_dur_us", 0)):
    n = cnt[key]
    g = c.get("GRBM_GUI_ACTIVE", 0) / n / 8.0  # the counter is reported per XCD (8 rows per dispatch): average them
    line = "%-58s grid=%-8s n=%-3d" % (key[0], key[1], n)
    if c["_dur_us"]:
        line += " dur=%8.1fus" % (c["_dur_us"] / n)
        # effective clock = GRBM_GUI_ACTIVE / duration: only meaningful where that counter was collected (the SQ pass, not the FETCH / WRITE
        # passes) and where the launch is long enough for the ratio not to be dominated by the counter's start / stop skew (>= 20 us)
        if g and c["_dur_us"] / n >= 20.0: line += " clk=%5.0fMHz" % (g / (c["_dur_us"] / n))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and g: line += " mfma_util=%5.1f%%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (g * 1024))
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c and c.get("SQ_WAVE_CYCLES"): line += " %s=%4.1f%%" % (k[3:], 100 * c[k] / c["SQ_WAVE_CYCLES"])
    for k in ("SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE"):
        if k in c: line += " %s=%.3g" % (k, c[k] / n)
    known = {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
             "SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "_dur_us"}
    for k in sorted(c):  # any other counter: per-dispatch average and its share of SQ_WAVE_CYCLES
        if k not in known:
            line += " %s=%.3g" % (k, c[k] / n)
            if c.get("SQ_WAVE_CYCLES"): line += "(%.1f%%)" % (100 * c[k] / c["SQ_WAVE_CYCLES"])
    print(line)
