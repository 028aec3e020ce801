"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per
launch (profiles/r01_pmc_traffic.json).  Usage (GPU box):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/f -o p -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/w -o p -- python bench.py ...
    python tools/pmc_traffic.py out/f out/w profiles/r01_pmc_traffic.json
Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled.
"""
import collections, csv, glob, json, sys


def load(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*counter_collection.csv"):
        rows = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            rows[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (_, k), v in rows.items():
            agg[k][0] += v
            agg[k][1] += 1
    return agg


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0]


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if "at::native" in k or "rocclr" in k:
        continue
    f, nf = fetch.get(k, [0, 1])
    w, nw = write.get(k, [0, 1])
    out[short(k)] = {"launches": int(max(nf, nw)),
                     "fetch_bytes_per_launch_raw": f / max(nf, 1) * 1024,
                     "read_bytes_per_launch_corrected": 2 * f / max(nf, 1) * 1024,
                     "write_bytes_per_launch": w / max(nw, 1) * 1024}
    out[short(k)]["hbm_bytes_per_launch"] = (out[short(k)]["read_bytes_per_launch_corrected"] +
                                             out[short(k)]["write_bytes_per_launch"])
# the kernel sources this profile was taken on (bench.py refuses a profile whose head differs from the run's)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["_meta"] = {"head": bench.kernel_tree_sha(),
                "what": "sha256/16 of centernet_amd/csrc/*.hip, *.h, include/centernet_amd.h, centernet_amd/engine.py"}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if k == "_meta":
        continue
    print("%-60s launches %4d  read %9.1f MB  write %9.1f MB" % (
        k[:60], v["launches"], v["read_bytes_per_launch_corrected"] / 1e6, v["write_bytes_per_launch"] / 1e6))
