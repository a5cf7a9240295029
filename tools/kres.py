#!/usr/bin/env python3
"""Per-kernel resource usage of one HIP source (SGPR/VGPR, spills, scratch, occupancy) from hipcc's remark pass.
    python tools/kres.py tinydiffusionmodels_amd/csrc/conv_s16.hip [substring]"""
import re, subprocess, sys
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", "include", "-I", "tinydiffusionmodels_amd/csrc",
       "-DNDEBUG", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    if pat in r["name"]:
        print(f'{r["name"][:100]:100s} sgpr {r.get("TotalSGPRs","?"):>4} vgpr {r.get("VGPRs","?"):>4} agpr {r.get("AGPRs","?"):>3} '
              f'spill s{r.get("SGPRs Spill","?")}/v{r.get("VGPRs Spill","?")} scratch {r.get("ScratchSize [bytes/lane]","?"):>4} occ {r.get("Occupancy [waves/SIMD]","?")}')
