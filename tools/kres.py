"""Dev tool: per-kernel resource table (VGPRs, AGPRs, scratch, LDS, occupancy) of one .hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: python tools/kres.py lq_mpc_amd/csrc/lqmpc_r16.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("lqmpc::", "")}
        rows.append(cur)
        continue
    m = re.search(r"\s(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split()[0]] = int(m.group(2))
print("%-60s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "LDS", "occ"))
for r in rows:
    print("%-60s %5d %5d %7d %6d %4d" % (r["name"][:60], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("LDS", -1), r.get("Occupancy", -1)))
