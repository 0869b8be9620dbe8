"""Synthetic inputs of bench.py and the tests: the extended-XYZ subset the fixtures use, GPUMD's `replicate` atom
order, seeded crystals of the three model families (PbTe, carbon, 16-metal alloy) and Maxwell velocities.  Plain
numpy; nothing here touches the oracle or the GPU."""
import os
import re

import numpy as np

K_B = 8.617343e-5
TIME_UNIT = 10.18051  # fs per natural time unit (src/utilities/common.cuh:26)
MASS = {"Te": 127.6, "Pb": 207.2, "C": 12.011, "Ba": 137.327, "Zr": 91.224, "O": 15.999, "H": 1.008, "Si": 28.085}
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def golden(*parts):
    return os.path.join(GOLDEN, *parts)


def read_xyz_frames(path):
    """extended XYZ (the subset of read_xyz.cu:141-400 the fixtures use)"""
    frames = []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines) and lines[i].strip():
        n = int(lines[i].split()[0])
        comment = lines[i + 1]
        kv = {}
        for m in re.finditer(r'(\w+)=("([^"]*)"|(\S+))', comment):
            kv[m.group(1).lower()] = m.group(3) if m.group(3) is not None else m.group(4)
        lat = np.array([float(x) for x in kv["lattice"].split()]).reshape(3, 3)  # rows a,b,c
        props = kv["properties"].split(":")
        cols = []
        off = 0
        for k in range(0, len(props), 3):
            name, typ, cnt = props[k].lower(), props[k + 1], int(props[k + 2])
            cols.append((name, typ, cnt, off))
            off += cnt
        body = [lines[i + 2 + a].split() for a in range(n)]
        fr = {"n": n, "lattice": lat, "comment": kv}
        # GPUMD stores h = [a b c] as columns: h[0]=ax h[1]=bx h[2]=cx h[3]=ay ... (read_xyz.cu:208-216)
        fr["h"] = np.ascontiguousarray(lat.T.reshape(9))
        pbc = kv.get("pbc", "T T T").split()
        fr["pbc"] = np.array([1 if p.upper().startswith("T") else 0 for p in pbc], dtype=np.int32)
        for name, typ, cnt, o in cols:
            if typ == "S":
                fr[name] = [b[o] for b in body]
            else:
                fr[name] = np.array([[float(x) for x in b[o:o + cnt]] for b in body])
        for key in ("energy",):
            if key in kv:
                fr[key] = float(kv[key])
        if "virial" in kv:
            fr["virial"] = np.array([float(x) for x in kv["virial"].split()])
        frames.append(fr)
        i += 2 + n
    return frames


def types_from_species(species, symbols):
    idx = {s: k for k, s in enumerate(symbols)}
    return np.array([idx[s] for s in species], dtype=np.int32)


def soa(pos_nx3):
    """(N,3) -> GPUMD SoA [x..|y..|z..]"""
    return np.ascontiguousarray(np.asarray(pos_nx3, dtype=np.float64).T.reshape(-1))


def replicate(h, species_or_type, pos_nx3, reps):
    """Supercell with GPUMD's atom order (replicate.cu:50-71): i, j, k outer loops, basis inner."""
    H = np.asarray(h, dtype=np.float64).reshape(3, 3)  # columns a,b,c
    a, b, c = H[:, 0], H[:, 1], H[:, 2]
    pos = np.asarray(pos_nx3, dtype=np.float64)
    out = []
    typ = []
    for i in range(reps[0]):
        for j in range(reps[1]):
            for k in range(reps[2]):
                out.append(pos + i * a + j * b + k * c)
                typ.append(np.asarray(species_or_type))
    Hn = H * np.array(reps, dtype=np.float64)[None, :]
    return np.ascontiguousarray(Hn.reshape(9)), np.concatenate(typ), np.concatenate(out)


def maxwell_velocities(mass, temperature, seed=11):
    """Gaussian velocities at `temperature`, zero net momentum, SoA [vx|vy|vz] in natural units."""
    rng = np.random.default_rng(seed)
    n = len(mass)
    v = rng.normal(0.0, 1.0, (3, n)) * np.sqrt(K_B * temperature / mass)[None, :]
    v -= (v * mass[None, :]).sum(axis=1, keepdims=True) / mass.sum()
    t_now = (mass[None, :] * v * v).sum() / (3.0 * n * K_B)
    v *= np.sqrt(temperature / t_now)
    return np.ascontiguousarray(v.reshape(-1))


# ---- crystals.  Positions are NOT wrapped into the cell: a rattled atom may sit a little outside it, which is what a
# ---- block of a larger (multi-GPU) crystal needs; Force::compute / the decomposition wrap them where they belong.
def pbte_block(reps, rattle=0.02, seed=42, temperature=300.0):
    """`replicate` of the 250-atom PbTe cell of examples/gpumd_static + rattle -> (h, type, x_soa, mass, vel)"""
    fr = read_xyz_frames(golden("PbTe", "model.xyz"))[0]
    typ0 = types_from_species(fr["species"], ["Te", "Pb"])
    h, typ, pos = replicate(fr["h"], typ0, fr["pos"], reps)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    typ = typ.astype(np.int32)
    mass = np.where(typ == 0, MASS["Te"], MASS["Pb"]).astype(np.float64)
    return h, typ, soa(pos), mass, maxwell_velocities(mass, temperature, seed=seed + 1)


def rocksalt_block(cells, a=6.5704, rattle=0.01, seed=42, temperature=300.0):
    """orthogonal rock-salt PbTe (SURVEY 8d.3 variant): 8 atoms per conventional cell"""
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5],
                      [.5, 0, 0], [0, .5, 0], [0, 0, .5], [.5, .5, .5]]) * a
    typ0 = np.array([1, 1, 1, 1, 0, 0, 0, 0], dtype=np.int32)  # Pb = 1, Te = 0 (nep4 2 Te Pb)
    h, typ, pos = replicate(np.diag([a, a, a]).reshape(9), typ0, basis, cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    typ = typ.astype(np.int32)
    mass = np.where(typ == 0, MASS["Te"], MASS["Pb"]).astype(np.float64)
    return h, typ, soa(pos), mass, maxwell_velocities(mass, temperature, seed=seed + 1)


def diamond_block(cells, a=3.57, rattle=0.02, seed=42, temperature=300.0, mass=None):
    fcc = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
    basis = np.concatenate([fcc, fcc + 0.25]) * a
    h, typ, pos = replicate(np.diag([a, a, a]).reshape(9), np.zeros(8, dtype=np.int32), basis, cells)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    m = np.full(len(typ), MASS["C"] if mass is None else mass)
    return h, typ.astype(np.int32), soa(pos), m, maxwell_velocities(m, temperature, seed=seed + 1)


def fcc_alloy_block(cells, a=3.9, num_types=16, rattle=0.02, seed=42, temperature=300.0, mass=100.0):
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    h, _, pos = replicate(np.diag([a, a, a]).reshape(9), np.zeros(4, dtype=np.int32), basis, cells)
    rng = np.random.default_rng(seed)
    typ = rng.integers(0, num_types, len(pos)).astype(np.int32)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    m = np.full(len(typ), float(mass))
    return h, typ, soa(pos), m, maxwell_velocities(m, temperature, seed=seed + 1)
