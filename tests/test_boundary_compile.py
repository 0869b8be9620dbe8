"""The reference-side binding of INTEGRATION.md section 1 is real code: its two listings (nep_mi.cuh / nep_mi.cu,
a `Potential` subclass) are extracted from the document and compiled with hipcc for gfx950 against the reference's
OWN headers (src/force/potential.cuh, src/utilities/gpu_vector.cuh, src/model/box.cuh) and include/nepmi.h; every
nepmi_* symbol the object needs is exported by libnepmi.so.  Runs where /root/reference exists (this container)."""
import os
import re
import shutil
import subprocess

import pytest

import helpers as H

REF_SRC = "/root/reference/src"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "force", "potential.cuh")) or not os.path.exists(HIPCC),
                    reason="needs the reference checkout and hipcc")
def test_adaptor_compiles_against_reference_headers(tmp_path):
    doc = open(os.path.join(H.ROOT, "INTEGRATION.md")).read()
    got = {}
    for block in re.findall(r"```cpp\n(.*?)```", doc, re.S):
        first = block.splitlines()[0].strip()
        if first in ("// nep_mi.cuh", "// nep_mi.cu"):
            got[first[3:]] = block
    assert set(got) == {"nep_mi.cuh", "nep_mi.cu"}
    for name, text in got.items():
        (tmp_path / name).write_text(text)
    obj = tmp_path / "nep_mi.o"
    cmd = [HIPCC, "-DUSE_HIP", "--offload-arch=gfx950", "-x", "hip", "-std=c++17", "-Wall", "-Werror",
           "-I" + REF_SRC, "-I" + os.path.join(REF_SRC, "force"), "-I" + os.path.join(H.ROOT, "include"),
           "-c", str(tmp_path / "nep_mi.cu"), "-o", str(obj)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    undefined = subprocess.run(["nm", "-u", str(obj)], capture_output=True, text=True).stdout
    wanted = sorted({l.split()[-1] for l in undefined.splitlines() if l.split() and l.split()[-1].startswith("nepmi_")})
    assert "nepmi_potential_compute" in wanted and "nepmi_engine_create" in wanted and "nepmi_model_load" in wanted
    lib = os.path.join(H.ROOT, "gpumd_amd", "lib", "libnepmi.so")
    if not os.path.exists(lib):
        lib = os.path.join(H.ROOT, "tests", "emu", "libnepmi_emu.so")    # same C ABI, host build
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    names = {l.split()[-1] for l in exported.splitlines() if l.split()}
    assert not [w for w in wanted if w not in names]
    # the object defines the subclass the factory hook constructs
    defined = subprocess.run(["nm", "-C", "--defined-only", str(obj)], capture_output=True, text=True).stdout
    assert "NEP_MI::compute(Box&" in defined and "NEP_MI::NEP_MI(char const*, int)" in defined
