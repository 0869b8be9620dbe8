"""TEST INFRASTRUCTURE.  Writes the two listings of INTEGRATION.md section 1 (nep_mi.cuh, nep_mi.cu -- the reference-side
binding a GPUMD maintainer would add) to a directory, verbatim: the document is the single source of the adaptor that
tests/test_boundary_compile.py compiles and that oracle/ref_gpumd.mk links into the reference's own `gpumd`."""
import os
import re
import sys

doc, out = sys.argv[1], sys.argv[2]
text = open(doc).read()
got = {}
for block in re.findall(r"```cpp\n(.*?)```", text, re.S):
    first = block.splitlines()[0].strip()
    if first in ("// nep_mi.cuh", "// nep_mi.cu"):
        got[first[3:]] = block
assert set(got) == {"nep_mi.cuh", "nep_mi.cu"}, sorted(got)
os.makedirs(out, exist_ok=True)
for name, body in got.items():
    with open(os.path.join(out, name), "w") as f:
        f.write(body)
