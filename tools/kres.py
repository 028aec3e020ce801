"""Per-kernel register / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python tools/kres.py centernet_amd/csrc/cn_conv3x3.hip [substring filter]
"""
import re, subprocess, sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + __import__("os").environ.get("KRES_FLAGS", "").split(), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.strip()] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    if flt and flt not in name:
        continue
    print("%-86s V %3s A %3s spill %s occ %s scratch %s" % (name[:86], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
          r.get("Occupancy [waves/SIMD]"), r.get("ScratchSize [bytes/lane]")))
