#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels in a gfx950 code object, from its metadata notes.
usage: kernel_resources.py gpumd_amd/lib/libnepmi.so [substring ...]"""
import re
import subprocess
import sys

lib = sys.argv[1]
subs = sys.argv[2:]
# the device code object is an offload bundle inside the host .so: unbundle first
import os, tempfile
tmp = tempfile.mkdtemp()
co = os.path.join(tmp, "dev.co")
r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + lib,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
if r.returncode != 0 or not os.path.exists(co):
    # a .so keeps the bundle in the .hip_fatbin section
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
cur = {}
rows = []
for line in notes.splitlines():
    m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and ("kernel" in v or "_Z" in v) and "cur_name" not in cur:
        pass
    if k in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "symbol"):
        cur[k] = v
    if k == "wavefront_size":
        if "symbol" in cur:
            rows.append(cur)
        cur = {}
for c in rows:
    name = subprocess.run(["c++filt", c["symbol"].replace(".kd", "").strip("'")], capture_output=True, text=True).stdout.strip()
    name = name.replace("nepmi::", "")
    if subs and not any(s in name for s in subs):
        continue
    print("%-4s vgpr %-4s agpr %-4s sgpr  scratch %-5s lds %-6s %s" % (c.get("vgpr_count"), c.get("agpr_count"), c.get("sgpr_count"),
          c.get("private_segment_fixed_size"), c.get("group_segment_fixed_size"), name[:150]))
