#!/usr/bin/env python3
"""Same-box comparison: the REFERENCE's own `gpumd` (its HIP build for gfx950: oracle/ref_gpumd.mk -> oracle/_ref/gpumd_ref,
-DDEBUG = fixed PRNG seed) and `gpumd-mi` on identical run.in / model.xyz / potential files, one after the other on
the same MI355X.  Measurement infrastructure: the reference binary is the comparator, never part of the product.

For every case: the "Speed of this run" line of both programs (atom*step/second over the `run` block, as the
reference defines it, src/main_gpumd/run.cu:324-326) and the row-by-row difference of thermo.out (same velocities
from the same rand() stream, so the trajectories coincide until chaos separates them).

    python profiles/ref_compare.py [--out gpurun_out/ref_compare] [--cases pbte_1m si_tersoff carbon_nvt unep ...]

Writes <out>/<case>/{ref,mi}/ (inputs, stdout, thermo.out) and <out>/summary.json.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "gpumd_ref")
MI = os.path.join(ROOT, "gpumd_amd", "bin", "gpumd-mi")
# the drop-in, executed: the reference's own host with its NEP factory line patched to the INTEGRATION.md adaptor, linked
# against libnepmi.so (oracle/ref_gpumd.mk: _ref/gpumd_ref_mi)
REF_MI = os.path.join(ROOT, "oracle", "_ref", "gpumd_ref_mi")
GOLD = os.path.join(ROOT, "tests", "golden")


def write_xyz(path, lattice9, species, pos_nx3):
    with open(path, "w") as f:
        f.write("%d\n" % len(species))
        f.write('pbc="T T T" Lattice="%s" Properties=species:S:1:pos:R:3\n' % " ".join("%.10g" % v for v in lattice9))
        for s, p in zip(species, pos_nx3):
            f.write("%s %.10f %.10f %.10f\n" % (s, p[0], p[1], p[2]))


def diamond_cell(a, symbol):
    fcc = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
    basis = np.concatenate([fcc, fcc + 0.25]) * a
    # a small deterministic displacement keeps the forces non-zero (a perfect lattice is a degenerate test)
    rng = np.random.default_rng(5)
    return [a, 0, 0, 0, a, 0, 0, 0, a], [symbol] * 8, basis + rng.normal(0.0, 0.01, basis.shape)


def fcc_alloy_cell(a, symbols, cells, seed=7):
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    pos = []
    for i in range(cells):
        for j in range(cells):
            for k in range(cells):
                pos.append(basis + np.array([i, j, k]) * a)
    pos = np.concatenate(pos)
    rng = np.random.default_rng(seed)
    spec = [symbols[t] for t in rng.integers(0, len(symbols), len(pos))]
    L = a * cells
    return [L, 0, 0, 0, L, 0, 0, 0, L], spec, pos + rng.normal(0.0, 0.02, pos.shape)


FINE = 0


def case_inputs(name, d):
    """Writes run.in, model.xyz and the potential file into d; returns the number of atoms after replication."""
    os.makedirs(d, exist_ok=True)
    if name in ("pbte_1m", "pbte_250k", "pbte_128k", "pbte_16k", "pbte_250"):
        reps = {"pbte_1m": 16, "pbte_250k": 10, "pbte_128k": 8, "pbte_16k": 4, "pbte_250": 1}[name]
        shutil.copy(os.path.join(GOLD, "PbTe", "model.xyz"), os.path.join(d, "model.xyz"))
        shutil.copy(os.path.join(GOLD, "PbTe", "nep.txt"), os.path.join(d, "nep.txt"))
        steps = 500 if reps >= 8 else 2000
        run = ("replicate %d %d %d\n" % (reps, reps, reps) if reps > 1 else "") + (
            "potential nep.txt\nvelocity 300\nensemble nve\ntime_step 1\ndump_thermo %d\nrun %d\n" % (steps // 10, steps))
        n = 250 * reps ** 3
    elif name == "pbte_temperature":
        # temperature-dependent NEP (nep4_temperature): a synthetic model (the PbTe file + one ANN input, made by
        # tests/test_temperature_nep.py) under a Berendsen ramp 300 -> 900 K: the ANN sees a new temperature every step
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pathlib
        from test_temperature_nep import make_temperature_model
        make_temperature_model(pathlib.Path(d), zbl=False, name="nep.txt")
        shutil.copy(os.path.join(GOLD, "PbTe", "model.xyz"), os.path.join(d, "model.xyz"))
        run = "replicate 4 4 4\npotential nep.txt\nvelocity 300\nensemble nvt_ber 300 900 100\ntime_step 1\ndump_thermo 20\nrun 200\n"
        n = 250 * 64
    elif name == "pbte_lmax8":
        # l_max_3body = 8 with the 222 and 1111 rows (80 sums per radial channel): a synthetic model on the PbTe descriptor
        # coefficients, made by tests/test_high_l.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pathlib
        from test_high_l import make_high_l
        make_high_l(pathlib.Path(d), 8, (1, 1), name="nep.txt")
        shutil.copy(os.path.join(GOLD, "PbTe", "model.xyz"), os.path.join(d, "model.xyz"))
        run = "replicate 4 4 4\npotential nep.txt\nvelocity 300\nensemble nve\ntime_step 1\ndump_thermo 20\nrun 200\n"
        n = 250 * 64
    elif name in ("pbte_lan", "pbte_bao"):
        # the stochastic thermostats (seeded from rand(): not reproducible in the reference's HIP build, compared statistically):
        # the reference's examples/gpumd_dynamic set-up on 16,000 atoms
        shutil.copy(os.path.join(GOLD, "PbTe", "model.xyz"), os.path.join(d, "model.xyz"))
        shutil.copy(os.path.join(GOLD, "PbTe", "nep.txt"), os.path.join(d, "nep.txt"))
        ens = "nvt_lan" if name == "pbte_lan" else "nvt_bao"
        run = "replicate 4 4 4\npotential nep.txt\nvelocity 300\nensemble %s 300 300 100\ntime_step 1\ndump_thermo 200\nrun 2000\n" % ens
        n = 250 * 64
    elif name == "si_tersoff":
        lat, spec, pos = diamond_cell(5.432, "Si")
        write_xyz(os.path.join(d, "model.xyz"), lat, spec, pos)
        shutil.copy(os.path.join(GOLD, "Si", "Si_Tersoff_1989.txt"), os.path.join(d, "potential.txt"))
        run = "replicate 12 12 12\npotential potential.txt\nvelocity 300\nensemble nve\ntime_step 1\ndump_thermo 1000\nrun 10000\n"
        n = 8 * 12 ** 3
    elif name in ("carbon_nvt", "carbon_nve", "carbon_nhc", "carbon_bdp"):
        lat, spec, pos = diamond_cell(3.57, "C")
        write_xyz(os.path.join(d, "model.xyz"), lat, spec, pos)
        shutil.copy(os.path.join(GOLD, "C", "nep.txt"), os.path.join(d, "nep.txt"))
        ens = {"carbon_nvt": "nvt_ber 300 300 100", "carbon_nhc": "nvt_nhc 300 300 100", "carbon_bdp": "nvt_bdp 300 300 100",
               "carbon_nve": "nve"}[name]
        run = "replicate 50 50 50\npotential nep.txt\nvelocity 300\nensemble %s\ntime_step 1\ndump_thermo 20\nrun 200\n" % ens
        n = 8 * 50 ** 3
    elif name == "unep":
        symbols = open(os.path.join(GOLD, "UNEP", "nep.txt")).readline().split()[2:]
        lat, spec, pos = fcc_alloy_cell(3.9, symbols, 16)
        write_xyz(os.path.join(d, "model.xyz"), lat, spec, pos)
        shutil.copy(os.path.join(GOLD, "UNEP", "nep.txt"), os.path.join(d, "nep.txt"))
        run = "replicate 4 4 4\npotential nep.txt\nvelocity 300\nensemble nve\ntime_step 1\ndump_thermo 20\nrun 200\n"
        n = 4 * 16 ** 3 * 64
    elif name == "unep_256k":
        # config 4's model where the run loop takes the many-type LDS scatter: 4 x 40^3 = 256,000 atoms = 1,000 bricks
        symbols = open(os.path.join(GOLD, "UNEP", "nep.txt")).readline().split()[2:]
        lat, spec, pos = fcc_alloy_cell(3.9, symbols, 40)
        write_xyz(os.path.join(d, "model.xyz"), lat, spec, pos)
        shutil.copy(os.path.join(GOLD, "UNEP", "nep.txt"), os.path.join(d, "nep.txt"))
        run = "potential nep.txt\nvelocity 2000\nensemble nve\ntime_step 1\ndump_thermo 10\nrun 100\n"
        n = 4 * 40 ** 3
    elif name == "carbon_262k":
        # config 5's model at >= 768 bricks: 8 x 32^3 = 262,144 atoms
        lat, spec, pos = diamond_cell(3.57, "C")
        write_xyz(os.path.join(d, "model.xyz"), lat, spec, pos)
        shutil.copy(os.path.join(GOLD, "C", "nep.txt"), os.path.join(d, "nep.txt"))
        run = "replicate 32 32 32\npotential nep.txt\nvelocity 2000\nensemble nve\ntime_step 1\ndump_thermo 10\nrun 100\n"
        n = 8 * 32 ** 3
    else:
        raise SystemExit("unknown case " + name)
    if FINE:
        # Row-by-row parity of the first steps: thermo every step, a short run, and the velocities written into
        # model.xyz (vel:R:3) so that neither program draws from rand() -- the ROCm runtime itself consumes draws of the
        # process-wide glibc stream, so the reference's HIP build is not reproducible run to run through `velocity`.
        run = re.sub(r"dump_thermo \d+", "dump_thermo 1", run)
        run = re.sub(r"\nrun \d+", "\nrun %d" % FINE, run)
        run = re.sub(r"velocity [^\n]*\n", "", run)
        m = re.search(r"replicate (\d+) (\d+) (\d+)\n", run)
        reps = tuple(int(v) for v in m.groups()) if m else (1, 1, 1)
        cap = 99 if name in ("unep_256k", "carbon_262k") else 10 if name == "pbte_250k" else 4 if name.startswith("pbte") else (20 if name.startswith("carbon") else (1 if name == "unep" else 99))
        reps = tuple(min(r, cap) for r in reps)  # explicit files: 13,824 to 64,000 atoms
        run = re.sub(r"replicate [^\n]*\n", "", run)
        from gpumd_amd import structures as S
        fr = S.read_xyz_frames(os.path.join(d, "model.xyz"))[0]
        h = np.asarray(fr["lattice"], dtype=np.float64).reshape(3, 3).T  # columns = a, b, c
        spec0, pos0 = list(fr["species"]), np.asarray(fr["pos"], dtype=np.float64)
        spec, pos = [], []
        for i in range(reps[0]):
            for j in range(reps[1]):
                for k in range(reps[2]):
                    spec += spec0
                    pos.append(pos0 + h @ np.array([i, j, k], dtype=np.float64))
        pos = np.concatenate(pos)
        lat = (h * np.asarray(reps, dtype=np.float64)[None, :]).T.reshape(9)
        mass = np.array([S.MASS.get(e, 100.0) for e in spec])  # only shapes the velocity distribution
        # (the two large cases start hot: perfect lattices at 300 K would not rebuild their lists within a hundred steps)
        # (diamond is stiff: only a very hot start moves an atom past skin / 2 within a hundred steps)
        t_start = {"unep_256k": 2000.0, "carbon_262k": 8000.0}.get(name, 300.0)
        vel = S.maxwell_velocities(mass, t_start, seed=3).reshape(3, -1).T / S.TIME_UNIT  # model.xyz carries A/fs
        with open(os.path.join(d, "model.xyz"), "w") as f:
            f.write("%d\n" % len(spec))
            f.write('pbc="T T T" Lattice="%s" Properties=species:S:1:pos:R:3:vel:R:3\n' % " ".join("%.12g" % v for v in lat))
            for e, p, v in zip(spec, pos, vel):
                f.write("%s %.12f %.12f %.12f %.15e %.15e %.15e\n" % (e, p[0], p[1], p[2], v[0], v[1], v[2]))
        n = len(spec)
    with open(os.path.join(d, "run.in"), "w") as f:
        f.write(run)
    return n


def run_binary(exe, d, timeout):
    t0 = time.time()
    try:
        # GPUMD_MI_DEBUG: gpumd-mi's counterpart of the reference's -DDEBUG build (fixed BDP noise seed)
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout,
                           env=dict(os.environ, GPUMD_MI_DEBUG="1"))
        out, rc = p.stdout.decode(errors="replace"), p.returncode
    except subprocess.TimeoutExpired as e:
        out, rc = (e.stdout or b"").decode(errors="replace") + "\n[timeout]\n", -9
    with open(os.path.join(d, "stdout.txt"), "w") as f:
        f.write(out)
    m = re.findall(r"Speed of this run = ([0-9.eE+-]+) atom\*step/second", out)
    t = re.findall(r"Time used for this run = ([0-9.eE+-]+) s", out)
    th = None
    p = os.path.join(d, "thermo.out")
    if os.path.exists(p):
        th = np.loadtxt(p, ndmin=2)
    return {"rc": rc, "wall_s": time.time() - t0, "speed": float(m[-1]) if m else None,
            "run_seconds": float(t[-1]) if t else None}, th


def compare_thermo(a, b):
    """Columns of thermo.out (dump_thermo.cu): T, K, U, six stress components, box.  Relative differences per row for
    T, K, U (U relative to |K| of the row: the physically relevant scale of an energy difference) and absolute (GPa)
    for the stresses."""
    if a is None or b is None or a.shape != b.shape:
        return {"comparable": False, "shape_ref": None if a is None else list(a.shape), "shape_mi": None if b is None else list(b.shape)}
    rows = []
    for r in range(a.shape[0]):
        K = abs(a[r, 1]) + 1e-300
        rows.append({"row": r, "T_ref": a[r, 0], "T_mi": b[r, 0], "dT_rel": abs(a[r, 0] - b[r, 0]) / (abs(a[r, 0]) + 1e-300),
                     "dK_rel": abs(a[r, 1] - b[r, 1]) / K, "dU_over_K": abs(a[r, 2] - b[r, 2]) / K,
                     "dU_rel": abs(a[r, 2] - b[r, 2]) / (abs(a[r, 2]) + 1e-300),
                     "dP_max_GPa": float(np.max(np.abs(a[r, 3:9] - b[r, 3:9])))})
    etot_ref = a[:, 1] + a[:, 2]
    etot_mi = b[:, 1] + b[:, 2]
    return {"comparable": True, "rows": rows,
            "first_row": rows[0], "last_row": rows[-1],
            "energy_drift_ref_rel": float((etot_ref[-1] - etot_ref[0]) / abs(etot_ref[0])),
            "energy_drift_mi_rel": float((etot_mi[-1] - etot_mi[0]) / abs(etot_mi[0]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_compare"))
    ap.add_argument("--cases", nargs="+", default=["pbte_1m", "si_tersoff", "carbon_nvt", "unep", "pbte_16k"])
    ap.add_argument("--timeout", type=float, default=240.0)
    ap.add_argument("--fine", type=int, default=0, help="thermo every step, run this many steps (MD-level parity)")
    ap.add_argument("--dropin", action="store_true",
                    help="also run oracle/_ref/gpumd_ref_mi (the reference host on libnepmi.so) on every case")
    args = ap.parse_args()
    global FINE
    FINE = args.fine
    for exe in (REF, MI):
        if not os.path.exists(exe):
            raise SystemExit("missing " + exe)
    summary = {}
    for name in args.cases:
        base = os.path.join(args.out, name)
        res = {}
        ths = {}
        binaries = [("ref", REF), ("mi", MI)]
        if args.dropin and os.path.exists(REF_MI) and name != "si_tersoff":
            binaries.append(("ref_mi", REF_MI))
        for tag, exe in binaries:
            d = os.path.join(base, tag)
            n = case_inputs(name, d)
            res[tag], ths[tag] = run_binary(exe, d, args.timeout)
            res["atoms"] = n
            # the replicated model and trajectory dumps are not needed afterwards
            for junk in ("model.xyz",):
                if os.path.getsize(os.path.join(d, junk)) > (1 << 20):
                    os.remove(os.path.join(d, junk))
        res["thermo"] = compare_thermo(ths["ref"], ths["mi"])
        if res["ref"]["speed"] and res["mi"]["speed"]:
            res["speedup_mi_over_ref"] = res["mi"]["speed"] / res["ref"]["speed"]
        if "ref_mi" in res:
            res["thermo_dropin_vs_ref"] = compare_thermo(ths["ref"], ths["ref_mi"])
            if res["ref"]["speed"] and res["ref_mi"]["speed"]:
                res["speedup_dropin_over_ref"] = res["ref_mi"]["speed"] / res["ref"]["speed"]
        summary[name] = res
        print(name, json.dumps({k: v for k, v in res.items() if not k.startswith("thermo")}), flush=True)
        if res["thermo"].get("comparable"):
            print("   thermo first row:", json.dumps(res["thermo"]["first_row"]))
            print("   thermo last row: ", json.dumps(res["thermo"]["last_row"]), flush=True)
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
