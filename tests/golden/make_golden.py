"""Regenerate tests/golden/ from the reference tree (run in the build container only; the GPU box
has no /root/reference).  Copies DATA fixtures the reference's own tests/examples hold for this
path (potential files, structures, known answers) -- no reference source code.

  python tests/golden/make_golden.py
"""
import os
import shutil

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

COPIES = {
    # PbTe (config 1 + 3): model, 250-atom cell, and the CUDA-path known answer
    "examples/nep_train/nep.txt": "PbTe/nep.txt",
    "examples/gpumd_static/model.xyz": "PbTe/model.xyz",
    "examples/gpumd_static/dump.xyz": "PbTe/dump.xyz",
    "examples/gpumd_static/run.in": "PbTe/run.in",
    # second PbTe model shape (n_max 4, basis 8)
    "tests/gpumd/dump_observer/PbTe_species/PbTe.txt": "PbTe/nep_B.txt",
    # BaZrO3 golden regression of tests_pytest
    "tests_pytest/fixtures/models/nep_BaZrO3.txt": "BaZrO3/nep.txt",
    "tests_pytest/fixtures/golden/bulk_bazro3.npz": "BaZrO3/bulk_bazro3.npz",
    "tests_pytest/fixtures/structures/BaZrO3-nat40-rattled.xyz": "BaZrO3/BaZrO3-nat40-rattled.xyz",
    # carbon + UNEP (configs 4, 5) and a nep3 model
    "potentials/nep/C_2022_NEP4.txt": "C/nep.txt",
    "tests/gpumd/dump_observer/carbon_observe/C_2022_NEP3.txt": "C/nep3.txt",
    "potentials/nep/Song-2024-UNEP-v1-AgAlAuCrCuMgMoNiPbPdPtTaTiVWZr.txt": "UNEP/nep.txt",
    "tests_pytest/fixtures/models/nep_water.txt": "water/nep.txt",
    "potentials/tersoff/Si_Tersoff_1989.txt": "Si/Si_Tersoff_1989.txt",
    # the remaining shipped NEP models: silicon with 3-, 4- and 5-body descriptors (l_max 4 0 0 / 4 2 0 /
    # 4 2 1) and the long-range carbon model (rc 7/4, MN 358/71)
    "potentials/nep/Si_2022_NEP4_3body.txt": "Si/nep_3body.txt",
    "potentials/nep/Si_2022_NEP4_4body.txt": "Si/nep_4body.txt",
    "potentials/nep/Si_2022_NEP4_5body.txt": "Si/nep_5body.txt",
    "potentials/nep/C_2024_NEP4.txt": "C/nep_2024.txt",
}


def main():
    for src, dst in COPIES.items():
        d = os.path.join(OUT, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(os.path.join(REF, src), d)
        os.chmod(d, 0o644)
    # nep_prediction known answers: all 25 frames (250 atoms each) of examples/nep_prediction
    pdir = os.path.join(OUT, "PbTe")
    shutil.copyfile(os.path.join(REF, "examples/nep_prediction/train.xyz"), os.path.join(pdir, "train_25frames.xyz"))
    os.chmod(os.path.join(pdir, "train_25frames.xyz"), 0o644)
    e = np.loadtxt(os.path.join(REF, "examples/nep_prediction/energy_train.out"))
    fo = np.loadtxt(os.path.join(REF, "examples/nep_prediction/force_train.out"))
    v = np.loadtxt(os.path.join(REF, "examples/nep_prediction/virial_train.out"))
    np.savez_compressed(os.path.join(pdir, "train_25frames_out.npz"), energy=e, force=fo, virial=v)
    # the reference's one trajectory-level golden: tests/gpumd/carbon (64,000-atom amorphous carbon, NEP4, 100 NVE steps,
    # thermo1.out from a DEBUG build = glibc's default rand() stream).  model.xyz (4 MB of text) is kept as the parsed
    # doubles; the test writes it back with %.17g, which reproduces the same bits.
    with open(os.path.join(REF, "tests/gpumd/carbon/model.xyz")) as f:
        lines = f.read().split("\n")
    n = int(lines[0])
    pos = np.array([[float(t) for t in ln.split()[1:4]] for ln in lines[2:2 + n]])
    assert all(ln.split()[0] == "C" for ln in lines[2:2 + n])
    os.makedirs(os.path.join(OUT, "C"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "C", "carbon_64000.npz"), pos=pos, header=np.array(lines[1]))
    shutil.copyfile(os.path.join(REF, "tests/gpumd/carbon/thermo1.out"), os.path.join(OUT, "C", "carbon_thermo1.out"))
    shutil.copyfile(os.path.join(REF, "tests/gpumd/carbon/run.in"), os.path.join(OUT, "C", "carbon_run.in"))
    for fn in ("carbon_thermo1.out", "carbon_run.in"):
        os.chmod(os.path.join(OUT, "C", fn), 0o644)
    angular_rows()
    print("golden fixtures written to", OUT)


def angular_rows(out=None):
    """Known answers of the reference's OWN find_q / accumulate_f12 (src/utilities/nep_utilities.cuh, compiled
    for the host by oracle/Makefile's _ref target) on seeded random input, for every invariant row including the
    extra 4-body ones (112/123/233/134) that no shipped model and no NEP_CPU covers."""
    import ctypes as C
    lib = C.CDLL(os.path.join(os.path.dirname(OUT), "..", "oracle", "_ref", "libnep_utils_ref.so"))
    nabc = lib.nepref_num_abc()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(20240924)
    flags = [[1, 1, 1, 1, 1, 1], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 1],
             [1, 0, 0, 1, 1, 0], [0, 1, 1, 0, 0, 1], [1, 1, 0, 0, 0, 0]]
    ncase, nA1 = 6, 3
    rec = dict(flags=np.array(flags, np.int32), s=[], Fp=[], r12=[], fn=[], fnp=[], q=[], f12=[])
    for c in range(ncase):
        s = np.zeros((nA1, nabc), np.float32)
        s[:, :24] = rng.normal(0, 0.7, (nA1, 24))
        Fp = rng.normal(0, 1, (10, nA1)).astype(np.float32)       # [row][n]
        r12 = rng.normal(0, 1.5, 3).astype(np.float32)
        fn, fnp = np.float32(rng.normal()), np.float32(rng.normal())
        d12 = np.float32(np.sqrt((r12.astype(np.float64) ** 2).sum()))
        qs, fs = [], []
        for fl in flags:
            num_L = 4 + sum(fl)
            q = np.zeros((10, nA1), np.float32)
            f = np.zeros((nA1, 3), np.float32)
            for n in range(nA1):
                lib.nepref_find_q(4, *fl, nA1, n, s[n].ctypes.data_as(fp), q.ctypes.data_as(fp))
                lib.nepref_accumulate_f12(4, *fl, num_L, n, nA1, C.c_float(d12), r12.ctypes.data_as(fp), C.c_float(fn),
                                          C.c_float(fnp), Fp.ctypes.data_as(fp), s.ctypes.data_as(fp),
                                          f[n].ctypes.data_as(fp))
            qs.append(q)
            fs.append(f)
        for k, v in (("s", s[:, :24]), ("Fp", Fp), ("r12", r12), ("fn", fn), ("fnp", fnp), ("q", qs), ("f12", fs)):
            rec[k].append(v)
    out = out or os.path.join(OUT, "rows", "angular_rows_ref.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez(out, **{k: np.array(v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
