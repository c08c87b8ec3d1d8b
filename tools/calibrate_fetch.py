"""FETCH_SIZE / WRITE_SIZE of rocprofv3 against KNOWN byte counts (tools/ubench/gather_fetch.hip), per access pattern.
usage: python tools/calibrate_fetch.py <true_bytes.json> <dir with the --pmc FETCH_SIZE run> <dir with the WRITE_SIZE run> <out.json>
The counters are reported in KILOBYTES by rocprofv3 (x 1024 here, like tools/summarize_profile.py)."""
import csv, glob, json, sys, collections

true = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
out = {"N_records": true["N"], "patterns": {}}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:4]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].strip()
            name = name.replace("void ", "")
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, tb in true["true_bytes"].items():
    v = vals.get(k, {})
    rec = {"true_read_bytes": tb["read"], "true_write_bytes": tb["write"]}
    if "read_sectors64" in tb:
        rec["true_read_bytes_whole_64B_sectors"] = tb["read_sectors64"]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if v.get(c):
            x = sorted(v[c])[len(v[c]) // 2] * 1024.0
            rec[c + "_bytes"] = x
            t = tb["read"] if c == "FETCH_SIZE" else tb["write"]
            if t:
                rec[c + "_over_true"] = x / t
            if c == "FETCH_SIZE" and "read_sectors64" in tb:
                rec["FETCH_SIZE_over_whole_sectors"] = x / tb["read_sectors64"]
    out["patterns"][k] = rec
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
