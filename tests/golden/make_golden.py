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
    # nep_prediction known answers: first 2 of the 25 frames (250 atoms each)
    pdir = os.path.join(OUT, "PbTe")
    with open(os.path.join(REF, "examples/nep_prediction/train.xyz")) as f:
        lines = f.readlines()
    n = int(lines[0].split()[0])
    with open(os.path.join(pdir, "train_2frames.xyz"), "w") as f:
        f.writelines(lines[: 2 * (n + 2)])
    e = np.loadtxt(os.path.join(REF, "examples/nep_prediction/energy_train.out"))[:2]
    fo = np.loadtxt(os.path.join(REF, "examples/nep_prediction/force_train.out"))[: 2 * n]
    v = np.loadtxt(os.path.join(REF, "examples/nep_prediction/virial_train.out"))[:2]
    np.savez(os.path.join(pdir, "train_2frames_out.npz"), energy=e, force=fo, virial=v)
    # first frame of the 64k-atom carbon test cell is too big to ship; keep its header only
    with open(os.path.join(REF, "tests/gpumd/carbon/model.xyz")) as f:
        head = [next(f) for _ in range(2)]
    os.makedirs(os.path.join(OUT, "C"), exist_ok=True)
    with open(os.path.join(OUT, "C", "carbon_64000_header.txt"), "w") as f:
        f.writelines(head)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
