#!/usr/bin/env python3
"""Builds a JIT core ahead of time into gpumd_amd/lib/jit/ (what capi_jit.h would compile at the first nepmi_model_load of a
model of that shape): python3 tools/build_jit_core.py n_r,k_r,n_a,k_a,n_L,types  -- or --model path/to/nep.txt"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.dirname(HERE)
FILES = ["engine.hip", "nep_model.cpp", "transport_tcp.cpp", "engine_impl.h", "capi_impl.h", "capi_jit.h", "capi_dispatch.inc",
         "dist_bodies.h", "dist_impl.h", "dist_capi_impl.h", "nep_dev.h", "nep_bodies.h", "nep_window.h", "nep_scatter.h",
         "nep_fused.h", "nep_highl.h", "nep_highl_tables.h", "nep_invariants_extra.h", "nep_md.h", "nep_model.h", "tersoff_bodies.h",
         "../../include/nepmi.h"]  # = capi_jit.h: source_files()


def source_hash():
    h = 1469598103934665603
    for f in FILES:
        for b in open(os.path.join(SRC, f), "rb").read():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def shape_of_model(path):
    tok = [ln.split() for ln in open(path).read().splitlines()[:8]]
    first = tok[0]
    ntypes = int(first[1])
    rows = {t[0]: t[1:] for t in tok[1:] if t}
    nr, na = (int(v) for v in rows["n_max"][:2])
    kr, ka = (int(v) for v in rows["basis_size"][:2])
    lm = [int(v) for v in rows["l_max"]]
    nl = lm[0] + (1 if len(lm) > 1 and lm[1] == 2 else 0) + (1 if len(lm) > 2 and lm[2] == 1 else 0)
    return (nr, kr, na, ka, nl, ntypes if ntypes <= 2 else 0)


def refresh():
    """rebuild, side by side, every core in gpumd_amd/lib/jit/ whose name carries another source hash (`make` calls this after
    libnepmi.so: a core is only ever loaded by a library built from the same sources)"""
    out_dir = os.path.join(SRC, "..", "lib", "jit")
    if not os.path.isdir(out_dir):
        return
    h = source_hash()
    shapes = set()
    for f in os.listdir(out_dir):
        if f.startswith("libnepmi_jit_") and f.endswith(".so"):
            parts = f[len("libnepmi_jit_"):-3].split("_")
            if len(parts) == 7 and parts[6] != h:
                shapes.add(",".join(parts[:6]))
    jobs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), sh], stdout=subprocess.DEVNULL) for sh in sorted(shapes)]
    for j in jobs:
        if j.wait() != 0:
            raise SystemExit("a JIT core did not build")
    if shapes:
        print("JIT cores rebuilt for the new sources:", ", ".join(sorted(shapes)))


def main():
    if sys.argv[1] == "--refresh":
        refresh()
        return
    if sys.argv[1] == "--hash":  # what the Makefile bakes into libnepmi.so (-DNEPMI_SRC_HASH): capi_jit.h
        print(source_hash())
        return
    if sys.argv[1] == "--model":
        shape = shape_of_model(sys.argv[2])
    else:
        shape = tuple(int(v) for v in sys.argv[1].split(","))
    out_dir = os.path.join(SRC, "..", "lib", "jit")
    os.makedirs(out_dir, exist_ok=True)
    name = "libnepmi_jit_%s_%s.so" % ("_".join(str(v) for v in shape), source_hash())
    out = os.path.join(out_dir, name)
    if os.path.exists(out):
        print(out, "(up to date)")
        return
    for f in os.listdir(out_dir):  # cores of older sources of this shape
        if f.startswith("libnepmi_jit_%s_" % "_".join(str(v) for v in shape)):
            os.remove(os.path.join(out_dir, f))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed", "-Wl,-Bsymbolic", "-Wl,-rpath,/opt/rocm/lib",
           "-DNEPMI_JIT_CORE", "-DNEPMI_JIT_SHAPE=" + ",".join(str(v) for v in shape), "-DNEPMI_SRC_HASH=0x%sull" % source_hash(), "-o", out,
           os.path.join(SRC, "engine.hip"), os.path.join(SRC, "nep_model.cpp"), os.path.join(SRC, "transport_tcp.cpp"), "-ldl"]
    subprocess.run(cmd, check=True)
    print(out)


if __name__ == "__main__":
    main()
