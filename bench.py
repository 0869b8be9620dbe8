#!/usr/bin/env python
"""bench.py -- atom-steps/s of the NEP NVE hot path (BASELINE.json's metric) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one MD step of Run::perform_a_run for `ensemble nve` (src/main_gpumd/run.cu:250-318):
velocity-Verlet half-kick + drift, pbc wrap, zero, NEP force call (skin check, list rebuild when
needed, descriptor, ANN, forces, virial), second half-kick -- over all atoms, through the C ABI of
libnepmi.so (nepmi_run_nve).  Workload at N = 1: BASELINE.json configs[2], PbTe 1,024,000 atoms
(`replicate 16 16 16` of the 250-atom cell of examples/gpumd_static, model examples/nep_train/nep.txt),
rattled, 300 K Maxwell velocities, dt = 1 fs.  Inputs are resident in HBM before the timed region.

One JSON line is printed by rank 0 (see the contract in the task description); it also carries
  "roofline":     dominant force kernel vs the HBM roofline, durations from HIP events recorded on
                  the engine's own stream inside the timed region
  "cpu_baseline": the reference's own NEP_CPU (oracle/_ref) timed on this box's host cores on a
                  bounded 16,000-atom sample of the same crystal (rank 0, N = 1 only)
"""
import argparse
import subprocess
import datetime
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = {"pbte": "atom-steps/sec, NEP PbTe NVE (nep4 2 Te Pb, examples/nep_train/nep.txt)",
          "pbte_ortho": "atom-steps/sec, NEP PbTe NVE, orthogonal rock-salt cell (examples/nep_train/nep.txt)",
          "carbon": "atom-steps/sec, NEP carbon (potentials/nep/C_2022_NEP4.txt)",
          "unep": "atom-steps/sec, NEP UNEP-v1 16-metal alloy (potentials/nep/Song-2024-UNEP-v1)",
          "carbon2024": "atom-steps/sec, NEP C_2024 (potentials/nep/C_2024_NEP4.txt), a model shape outside the compiled set",
          "si_tersoff": "atom-steps/sec, Tersoff-1989 Si NVE (BASELINE config 2: examples/gpumd_benchmark/Si_Tersoff)"}
KERNEL_NAMES = ["gather_skin_check", "radial_descriptor", "angular_descriptor", "ann", "angular_partial_force",
                "force_assemble", "velocity_verlet", "list_rebuild"]


def _oracle_helpers():
    """tests/helpers.py holds the loaders of the CPU checkers (oracle/, oracle/_ref): only the cpu_baseline leg may
    touch them."""
    tests = os.path.join(ROOT, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    import helpers
    return helpers


def build_workload(name, reps, seed):
    """-> (label, nep.txt, h, type, x_soa, mass, vel).  `pbte` is BASELINE config 3 (the bench line);
    `pbte_ortho` its orthogonal-cell variant; `carbon` (C_2022_NEP4, diamond) and `unep` (UNEP-v1,
    16-metal fcc alloy) are the model families of configs 5 and 4."""
    from gpumd_amd import structures as S
    if name == "pbte":
        h, typ, x, mass, vel = S.pbte_block(reps, seed=seed)
        return ("PbTe %d atoms (replicate %d %d %d of the 250-atom cell), NEP, dt 1 fs, velocities drawn at 300 K "
                "(the hot model.xyz snapshot equilibrates near 570 K)" % ((len(typ),) + tuple(reps)),
                S.golden("PbTe", "nep.txt"), h, typ, x, mass, vel)
    if name == "pbte_ortho":
        cells = (int(2.5 * reps[0]), int(2.5 * reps[1]), 5 * reps[2])  # 16 16 16 -> 40 x 40 x 80 cells = 1,024,000 atoms
        h, typ, x, mass, vel = S.rocksalt_block(cells, seed=seed)
        return ("PbTe rock-salt %d atoms (%dx%dx%d conventional cells, orthogonal), NEP, dt 1 fs, 300 K" % ((len(typ),) + cells),
                S.golden("PbTe", "nep.txt"), h, typ, x, mass, vel)
    if name == "carbon":
        cells = tuple(5 * r for r in reps)  # 16 16 16 -> 80^3 diamond cells = 4,096,000 atoms; use --reps 10 10 10 for 1 M
        h, typ, x, mass, vel = S.diamond_block(cells, seed=seed)
        return ("diamond C %d atoms (%dx%dx%d cells), C_2022_NEP4, 300 K" % ((len(typ),) + cells),
                S.golden("C", "nep.txt"), h, typ, x, mass, vel)
    if name == "carbon2024":
        # a shipped potential OUTSIDE the library's compiled shapes (n_max 12 8, basis_size 16 12, rc 7 / 4 A, 100 neurons):
        # served by a JIT core (capi_jit.h) or, with NEPMI_JIT=0, by the run-time-shape kernels
        cells = tuple(4 * r for r in reps)  # 10 10 10 -> 40^3 diamond cells = 512,000 atoms
        h, typ, x, mass, vel = S.diamond_block(cells, seed=seed)
        return ("diamond C %d atoms (%dx%dx%d cells), C_2024_NEP4 (potentials/nep), 300 K" % ((len(typ),) + cells),
                S.golden("C", "nep_2024.txt"), h, typ, x, mass, vel)
    if name == "unep":
        cells = tuple(4 * r for r in reps)
        h, typ, x, mass, vel = S.fcc_alloy_block(cells, seed=seed)
        return ("fcc 16-metal alloy %d atoms (%dx%dx%d cells), UNEP-v1 + ZBL, 300 K" % ((len(typ),) + cells),
                S.golden("UNEP", "nep.txt"), h, typ, x, mass, vel)
    if name == "si_tersoff":
        cells = tuple(max(1, (3 * r) // 4) for r in reps)  # 16 16 16 -> 12 12 12 diamond cells = 13,824 atoms (config 2)
        h, typ, x, mass, vel = S.diamond_block(cells, a=5.432, rattle=0.0, seed=seed, mass=28.085)
        return ("diamond Si %d atoms (%dx%dx%d cells, a = 5.432 A), Tersoff-1989, dt 1 fs, 300 K" % ((len(typ),) + cells),
                S.golden("Si", "Si_Tersoff_1989.txt"), h, typ, x, mass, vel)
    raise SystemExit("unknown workload " + name)


def algorithmic_bytes(info, nn_r, nn_a):
    """compulsory HBM bytes per atom-step, split per kernel."""
    dim, nr1 = info.dim, info.n_max_radial + 1
    per_kernel = {
        "velocity_verlet": 128.0 + 80.0,                        # VV1 (m, f, x rw, v rw) + VV2 (m, f, v rw)
        "gather_skin_check": 24.0,                              # skin check re-reads x
        "radial_descriptor": 28.0 + 4.0 * nn_r + 4.0 * nr1,      # x+type, radial list, q_radial out
        "angular_descriptor": 28.0 + 4.0 * nn_a + 4.0 * (dim - nr1),
        "ann": 4.0 * dim + 4.0 * dim + 8.0,                      # q in, Fp out, pe
        "angular_partial_force": 4.0 * nn_a + 4.0 * (dim - nr1) + 12.0 * nn_a,   # list, Fp in, f12 out
        "force_assemble": 28.0 + 4.0 * nn_r + 4.0 * nr1 + 12.0 * nn_a + 24.0 + 72.0,  # x, list, Fp, f12 in; f, virial out
    }
    total = 232.0 + 160.0 + 8.0 * (nn_r + nn_a) + 8.0 * dim + 24.0 * nn_a
    return per_kernel, total


def fused_angular_own_bytes(info, nn_a):
    """What the ONE-kernel form of angular descriptor + ANN + partial forces (nep_fused.h) has to move itself: x + type, the angular
    list, the radial part of q in; pe, the radial part of Fp (the force assembly's input) and f12 out.  The q / Fp / s round trips
    between the three stages of the SURVEY 8(d) split are not compulsory for it -- they stay on-chip."""
    nr1 = info.n_max_radial + 1
    return 28.0 + 4.0 * nn_a + 4.0 * nr1 + 8.0 + 4.0 * nr1 + 12.0 * nn_a


def tersoff_bytes(nn):
    """Tersoff-1989: compulsory HBM bytes per atom-step with FP64 throughout.  The two
    force kernels occupy the engine's radial and force-assembly slots."""
    per_kernel = {
        "velocity_verlet": 128.0 + 80.0,
        "gather_skin_check": 24.0,
        "radial_descriptor": 28.0 + 4.0 * nn + 8.0 + 40.0 * nn,        # x+type, list; pe, (b, b', f12[3]) per bond out
        "force_assemble": 28.0 + 4.0 * nn + 2 * 24.0 * nn + 24.0 + 72.0,  # x, list, f12 own + gathered reverse; f, virial
    }
    total = 232.0 + 160.0 + 8.0 * nn + 88.0 * nn
    return per_kernel, total


def own_flops(info, nn_cand, nn_r, nn_a):
    """FP32 operations per atom-step the ENGINE executes (fma = 2), counted from the kernel bodies for the model's
    shape (DESIGN.md section 5 lists the terms): not the reference's count (8e4 for PbTe)."""
    kr, nr1 = info.basis_size_radial + 1, info.n_max_radial + 1
    ka, na1 = info.basis_size_angular + 1, info.n_max_angular + 1
    T, dim, nneu = info.num_types, info.dim, info.num_neurons
    ts = T if T <= 2 else 1
    poly1, poly2 = 18.0, 34.0   # envelope: sine polynomial / sine + cosine polynomials
    radial = nn_cand * (12.0 + 3.0 + poly1 + (5.0 + 2.0 * 2 * (kr - 2) + 2.0 * kr) + 2.0 * kr * ts) + 2.0 * nr1 * kr * ts
    force = nn_r * (12.0 + 4.0 + poly2 + (8.0 + 10.0 * (kr - 2) + 6.0) + 2.0 * 2.0 * kr + 27.0)
    harm = 45.0
    adesc = nn_a * (10.0 + poly1 + 4.0 * ka + harm + 2.0 * na1 * ka + 2.0 * 24.0 * na1) + na1 * 160.0
    ann = 2.0 * nneu * dim + 2.0 * nneu * (dim + T * ((kr + 3) // 4 * 4)) + 12.0 * nneu
    recompute = adesc - na1 * 160.0 if info.MN_angular <= 16 else 0.0
    aforce = recompute + na1 * 200.0 + nn_a * (10.0 + poly2 + 10.0 * ka + 2.0 * 2.0 * na1 * ka + 2.0 * 2.0 * 24.0 * na1 + 160.0 + 20.0)
    return {"radial_descriptor": radial, "force_assemble": force, "angular_descriptor": adesc, "ann": ann,
            "angular_partial_force": aforce, "total": radial + force + adesc + ann + aforce}


class _stdout_to_stderr:
    """The reference's NEP_CPU prints its model summary with printf; keep this process's stdout for
    the one JSON line by pointing fd 1 at stderr while the CPU baseline runs."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)  # C stdio buffers of the checker library
        finally:
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def _cpu_loop(reps, seconds, max_calls):
    """NEP_CPU (or the C oracle when oracle/_ref is absent) stepping a PbTe replica: one iteration =
    compute() + a host velocity-Verlet update, i.e. one atom-STEP per atom -> (n, calls, seconds, kind)."""
    H = _oracle_helpers()
    from gpumd_amd import structures as S
    nep = S.golden("PbTe", "nep.txt")
    h, typ, x, mass, vel = S.pbte_block(reps)
    n = len(typ)
    if H.ref_available():
        eng, kind = H.RefNepCpu(nep), "reference"
        compute = lambda xx: eng.compute(typ, h, xx)
    else:
        eng, kind = H.Oracle(nep), "port"
        compute = lambda xx: eng.compute(typ, h, xx, precision=64, path=0)
    dt = 1.0 / H.TIME_UNIT
    x = H.oracle_apply_pbc(h, x)
    _, f, _ = compute(x)  # warm-up + initial force
    minv = np.tile(1.0 / mass, 3)
    calls, t0 = 0, time.perf_counter()
    while True:
        vel += 0.5 * dt * f * minv
        x = H.oracle_apply_pbc(h, x + dt * vel)
        _, f, _ = compute(x)
        vel += 0.5 * dt * f * minv
        calls += 1
        el = time.perf_counter() - t0
        if el > seconds or calls >= max_calls:
            break
    return n, calls, el, kind


def cpu_worker(seconds):
    """One of the P independent single-thread instances of the all-cores aggregate:
    a 2,000-atom replica (NEP_CPU allocates ~224 KB of neighbour tables per atom)."""
    n, calls, el, kind = _cpu_loop((2, 2, 2), seconds, 10 ** 9)
    sys.stderr.write("CPUWORKER %d %d %.6f %s\n" % (n, calls, el, kind))


def cpu_aggregate(seconds):
    """P independent NEP_CPU instances, one thread each, every instance on its own replica: the fair
    'all host cores' figure next to the stock (effectively serial) one."""
    import subprocess
    ncpu = os.cpu_count() or 1
    P = max(1, min(64, ncpu // 2))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds)], env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for _ in range(P)]
    t0 = time.perf_counter()
    atoms_steps, done, n0 = 0.0, 0, 0
    for pr in procs:
        err = pr.communicate()[1]
        for line in err.splitlines():
            if line.startswith("CPUWORKER"):
                _, n, calls, el, _kind = line.split()
                atoms_steps += int(n) * int(calls) / float(el)
                n0 = int(n)
                done += 1
    wall = time.perf_counter() - t0
    if done == 0:
        return None
    return {"value": atoms_steps, "unit": "atom-steps/s", "instances": done, "threads_per_instance": 1,
            "sample": "%d independent NEP_CPU instances, PbTe %d atoms each, %.0f s of stepping per instance "
                      "(%.0f s wall incl. start-up)" % (done, n0, seconds, wall)}


def cpu_baseline(seconds=12.0):
    """NEP_CPU (reference, compiled in place into oracle/_ref) on a 16,000-atom PbTe replica; one
    iteration = compute() + a host velocity-Verlet update, so that it is an atom-STEP."""
    n, calls, el, kind = _cpu_loop((4, 4, 4), seconds, 50)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)) if kind == "reference" else 1
    out = {"value": n * calls / el, "unit": "atom-steps/s", "cores": cores, "kind": kind,
           "sample": "PbTe %d atoms (replicate 4 4 4), %d NVE steps, %.1f s; NEP_CPU is serial outside its descriptor loop"
                     % (n, calls, el)}
    agg = cpu_aggregate(max(4.0, 0.75 * seconds))
    if agg:
        out["all_cores_aggregate"] = agg
    return out


def reference_gpu_baseline(steps=200, timeout=180.0):
    """The reference's OWN `gpumd` (its HIP build compiled for gfx950 from /root/reference/src by oracle/ref_gpumd.mk ->
    oracle/_ref/gpumd_ref, a prebuilt comparator that travels with the repository) on the bench workload itself
    (model.xyz of the 250-atom PbTe cell, `replicate 16 16 16`, NVE, dt 1 fs) on THIS GPU, after the timed region: the
    "Speed of this run" line it prints (src/main_gpumd/run.cu:324-326).  Part of the baseline leg: never the thing
    measured.  None when the binary is absent."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "gpumd_ref")
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import ref_compare as R
    from gpumd_amd import structures as S
    traj = None
    with tempfile.TemporaryDirectory() as d:
        n = R.case_inputs("pbte_1m", d)
        with open(os.path.join(d, "run.in")) as f:
            run = f.read()
        # velocities in model.xyz (vel:R:3 of the 250-atom cell; `replicate` repeats them with the cell, replicate.cu:51-72), no
        # draw: the reference's HIP build draws from the process-wide rand() stream the runtime also uses, so only this makes the
        # two programs start from the same state -- and lets their trajectories be compared below
        fr = S.read_xyz_frames(os.path.join(d, "model.xyz"))[0]
        spec, pos = list(fr["species"]), np.asarray(fr["pos"], dtype=np.float64)
        mass = np.array([S.MASS.get(e, 100.0) for e in spec])
        vel = S.maxwell_velocities(mass, 300.0, seed=3).reshape(3, -1).T / S.TIME_UNIT
        lat = np.asarray(fr["lattice"], dtype=np.float64).reshape(9)
        with open(os.path.join(d, "model.xyz"), "w") as f:
            f.write("%d\n" % len(spec))
            f.write('pbc="T T T" Lattice="%s" Properties=species:S:1:pos:R:3:vel:R:3\n' % " ".join("%.12g" % v for v in lat))
            for e, p, v in zip(spec, pos, vel):
                f.write("%s %.12f %.12f %.12f %.15e %.15e %.15e\n" % (e, p[0], p[1], p[2], v[0], v[1], v[2]))
        # (the `velocity 300` line stays: with velocities in the file the keyword draws nothing, velocity.cu:322, but it is what
        # uploads the REPLICATED host velocities -- `replicate` re-allocates the device array and does not fill it, run.cu:356-358)
        with open(os.path.join(d, "run.in"), "w") as f:
            f.write(re.sub(r"\nrun \d+", "\nrun %d" % steps, re.sub(r"dump_thermo \d+", "dump_thermo %d" % steps, run)))
        res, th_ref = R.run_binary(exe, d, timeout)
        # the same inputs through this repository's own host (gpumd-mi): the last thermo row of both, i.e. the bench path -- the
        # fused run loop in its scatter form at 1,024,000 atoms, list rebuilds included -- against the reference's trajectory
        if os.path.exists(R.MI) and th_ref is not None:
            for fn in ("thermo.out", "neighbor.out"):
                if os.path.exists(os.path.join(d, fn)):
                    os.remove(os.path.join(d, fn))
            res_mi, th_mi = R.run_binary(R.MI, d, timeout)
            if th_mi is not None and th_mi.shape == th_ref.shape:
                a, b = th_ref[-1], th_mi[-1]
                traj = {"steps": steps, "atoms": n,
                        "rel_dT": float(abs(b[0] / a[0] - 1.0)), "rel_dU": float(abs(b[2] / a[2] - 1.0)),
                        "max_abs_dP_GPa": float(np.abs(b[3:9] - a[3:9]).max()),
                        "gpumd_mi_speed": res_mi.get("speed"),
                        "note": "last thermo.out row of gpumd-mi (this engine: fused run loop, scatter-form assembly, list rebuilds "
                                "inside) vs the reference's gpumd on identical run.in / model.xyz (velocities from the file)"}
    if not res.get("speed"):
        return None
    return {"value": res["speed"], "unit": "atom-steps/s", "kind": "reference gpumd (src/makefile.hip flags, gfx950), same GPU",
            "sample": "PbTe %d atoms (replicate 16 16 16), %d NVE steps, %.2f s in its run block" % (n, steps, res["run_seconds"]),
            "trajectory_vs_this_engine": traj}


def cpu_baseline_tersoff(pot, h, typ, x, mass, vel, seconds=12.0):
    """The C restatement of tersoff1989.cu (oracle/tersoff_oracle.c, pinned against the reference's own kernels by
    tests/test_tersoff.py) stepping the SAME 13,824-atom Si system: one iteration = compute() + host velocity-Verlet."""
    H = _oracle_helpers()
    o = H.TersoffOracle(pot)
    n = len(typ)
    dt = 1.0 / H.TIME_UNIT
    x = H.oracle_apply_pbc(h, x.copy())
    vel = vel.copy()
    _, f, _ = o.compute(typ, h, x)
    minv = np.tile(1.0 / mass, 3)
    calls, t0 = 0, time.perf_counter()
    while True:
        vel += 0.5 * dt * f * minv
        x = H.oracle_apply_pbc(h, x + dt * vel)
        _, f, _ = o.compute(typ, h, x)
        vel += 0.5 * dt * f * minv
        calls += 1
        el = time.perf_counter() - t0
        if el > seconds or calls >= 2000:
            break
    return {"value": n * calls / el, "unit": "atom-steps/s", "cores": 1, "kind": "port",
            "sample": "Si %d atoms (the bench system itself), %d NVE steps of the C Tersoff oracle incl. its O(N) cell-list "
                      "neighbour search every call, %.1f s" % (n, calls, el)}


def load_traffic(n_atoms):
    """profiles/traffic_latest.json: the builder's counter passes, one entry per workload size (`by_atoms`), or the single
    entry of the earlier rounds' format"""
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(tfile):
        return {}
    tj = json.load(open(tfile))
    by = tj.get("by_atoms")
    if by:
        e = by.get(str(int(n_atoms)))
        return dict(e, source=e.get("source", tj.get("source"))) if e else {}
    return tj


def kernel_report(st, info, n_atoms_per_launch, st_all=None, tersoff=False, fused_angular=False, brick_force=False):
    """per-kernel mean durations (HIP events) + the roofline object of the dominant force kernel.
    st: stats of the timed region (timing mode 2: only the force-assembly slot is filled); st_all: stats of the
    instrumented pass after it (every slot) -- the timed region's own figure wins where both exist."""
    if tersoff:
        per_kernel, b_step = tersoff_bytes(st.mean_nn_radial)
    else:
        per_kernel, b_step = algorithmic_bytes(info, st.mean_nn_radial, st.mean_nn_angular)
    kern = {}
    if fused_angular and not tersoff:
        # nep_fused.h: descriptor + ANN + partial angular forces are ONE launch (timed in the angular-descriptor slot); its
        # algorithmic bytes are the three stages' shares of the SURVEY 8(d) split (the step total is unchanged)
        per_kernel = dict(per_kernel)
        per_kernel["angular_fused"] = per_kernel.pop("angular_descriptor") + per_kernel.pop("ann") + per_kernel.pop("angular_partial_force")
        own_fused = fused_angular_own_bytes(info, st.mean_nn_angular)
    if brick_force and not tersoff:
        # nep_brick.h: ... and the scatter-form force assembly in the same launch (angular slot); the fold in the force slot
        per_kernel["brick_force"] = per_kernel.pop("angular_fused") + per_kernel["force_assemble"] - 24.0
        per_kernel["force_fold"] = 24.0  # the forces written
        per_kernel.pop("force_assemble")
    for src in (st_all, st):
        if src is None:
            continue
        for k, name in enumerate(KERNEL_NAMES):
            if src.launches[k] > 0 and src.ms_kernel_sum[k] > 0.0:
                # launches of speculatively enqueued steps that were re-run after a list rebuild returned at once:
                # they are not work, the mean is over the launches that ran
                ran = int(src.launches[k]) - (int(src.discarded_steps) if 1 <= k <= 6 else 0)
                ran = max(ran, 1)
                if fused_angular and name == "angular_descriptor":
                    name = "brick_force" if brick_force else "angular_fused"
                if brick_force and name == "force_assemble":
                    name = "force_fold"
                kern[name] = {"launches": ran, "avg_ms": src.ms_kernel_sum[k] / ran, "slot": k,
                              "timed_in": "timed region" if src is st else "instrumented pass after the clock"}
    # every kernel priced the same way as the roofline object below: algorithmic bytes / duration / HBM peak, and (where the
    # builder's PMC pass matches this workload size) the counter traffic beside it
    tj_all = load_traffic(n_atoms_per_launch)
    for name, e in kern.items():
        if name in per_kernel and e["avg_ms"] > 0.0:
            e["algorithmic_bytes_per_atom"] = per_kernel[name]
            e["frac"] = per_kernel[name] * n_atoms_per_launch / (e["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if name in ("angular_fused", "brick_force") and fused_angular and not tersoff:
                # the fused kernel priced with ITS OWN compulsory bytes; the figure above uses the three stages' shares of the split
                e["frac_survey_split"] = e["frac"]
                e["own_bytes_per_atom"] = own_fused
                e["frac"] = own_fused * n_atoms_per_launch / (e["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if tj_all.get("atoms") == n_atoms_per_launch and name in tj_all.get("kernels", {}):
                e["traffic"] = tj_all["kernels"][name]["hbm_bytes_per_launch"]
    force_kernels = [k for k in kern if k in per_kernel and k not in ("velocity_verlet", "gather_skin_check")]
    ranked = sorted(force_kernels, key=lambda k: -kern[k]["avg_ms"])
    dom = ranked[0] if ranked else None
    roofline = None
    if dom:
        # HBM-side bytes per launch from a separate rocprofv3 PMC pass of this same command (profiles/*_pmc_*.csv); only
        # meaningful for the workload (number of atoms) it was taken on
        traffic, tj = None, load_traffic(n_atoms_per_launch)
        if tj.get("atoms") == n_atoms_per_launch and dom in tj.get("kernels", {}):
            traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
        dom_bytes = kern[dom].get("own_bytes_per_atom", per_kernel[dom])
        achieved = dom_bytes * n_atoms_per_launch / (kern[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": (tj.get("source", "profiles/traffic_latest.json") + " (builder's rocprofv3 PMC pass, replayed "
                                       "here: not measured in this run)") if traffic is not None else None,
                    "algorithmic_bytes_per_launch": dom_bytes * n_atoms_per_launch,
                    "algorithmic_bytes_per_atom": dom_bytes, "avg_launch_ms": kern[dom]["avg_ms"],
                    "note": "FP32-VALU/gather bound stage (SURVEY.md 8d); see step_hbm_frac for the whole step"}
        if dom in ("angular_fused", "brick_force"):
            roofline["frac_survey_split"] = kern[dom].get("frac_survey_split")
            roofline["note"] += ("; this launch is the angular descriptor, the ANN and the partial angular forces in one kernel, priced with its OWN "
                                 "compulsory bytes (x, type, angular list, radial q in; pe, radial Fp, f12 out): `frac_survey_split` is the same time "
                                 "priced with the three stages' shares of the SURVEY 8(d) split, which counts the q / Fp round trips the kernel keeps "
                                 "on-chip -- the kernel is vector-issue bound (fp32_valu below), not HBM bound")
        if len(ranked) > 1:  # the runner-up, priced the same way (two kernels can be tied to a per cent)
            k2 = ranked[1]
            roofline["second"] = {"kernel": k2, "avg_launch_ms": kern[k2]["avg_ms"], "frac": kern[k2].get("frac"),
                                  "algorithmic_bytes_per_atom": per_kernel[k2], "traffic": kern[k2].get("traffic")}
    return kern, roofline, b_step


def dominant_slot(eng, run_steps, probe_steps=4):
    """A short instrumented pass BEFORE the clock: which per-step force kernel (timing slots 1..5) takes longest?  Its slot
    is the one that carries HIP events inside the timed region (nepmi_engine_set_timing(16 + slot))."""
    eng.set_timing(1)
    run_steps(probe_steps)
    st = eng.stats(with_lists=False)
    eng.set_timing(0)
    best, slot = -1.0, 5
    for k in range(1, 6):
        if st.launches[k] > 0:
            ran = max(int(st.launches[k]) - int(st.discarded_steps), 1)
            if st.ms_kernel_sum[k] / ran > best:
                best, slot = st.ms_kernel_sum[k] / ran, k
    return slot


def gpu_state():
    """clocks and power state of the device the line was measured on (rocm-smi), to tell boxes apart: the same library runs
    the radial pass at 0.365 ms on one box and 0.405 ms on another (README round 4)"""
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        card = j.get("card0", next(iter(j.values())) if j else {})
        for k, v in card.items():
            if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power", "performance")):
                out[k] = v
    except Exception as e:  # never lose the bench line to this
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def measure_extra(workload, reps, steps, warmup, dev, generic=False):
    """One more measurement AFTER the clock of the bench line has stopped (config.extra_measurements): its own engine, the
    same protocol (inputs resident, warm-up, K steps between synchronisations, HIP events on the dominant kernel inside
    the timed region, an instrumented pass after it).  -> dict with value / ms_per_step / roofline."""
    import torch
    import gpumd_amd
    from gpumd_amd import structures as H
    label, nep_txt, h, typ, x, mass, vel = build_workload(workload, reps, 42)
    n = len(typ)
    model = gpumd_amd.Model(nep_txt)
    eng = gpumd_amd.NEP(model, n)
    if generic:
        eng.set_generic(True)
    dt = 1.0 / H.TIME_UNIT
    t_type, t_mass = torch.from_numpy(typ).to(dev), torch.from_numpy(mass).to(dev)
    t_x, t_v = torch.from_numpy(x).to(dev), torch.from_numpy(vel).to(dev)
    t_pe, t_f, t_w = (torch.zeros(k * n, dtype=torch.float64, device=dev) for k in (1, 3, 9))
    eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w)
    if warmup > 0:
        eng.run_nve(h, t_type, t_mass, dt, warmup, t_x, t_v, t_pe, t_f, t_w)
    eng.set_timing(16 + dominant_slot(eng, lambda k: eng.run_nve(h, t_type, t_mass, dt, k, t_x, t_v, t_pe, t_f, t_w, thermo_every=k)))
    reb0 = eng.stats().num_rebuild
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_nve(h, t_type, t_mass, dt, steps, t_x, t_v, t_pe, t_f, t_w, thermo_every=steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = eng.stats(with_lists=True)
    eng.set_timing(1)
    extra = max(10, min(steps, 40))
    eng.run_nve(h, t_type, t_mass, dt, extra, t_x, t_v, t_pe, t_f, t_w, thermo_every=extra)
    st_all = eng.stats(with_lists=False)
    eng.set_timing(0)
    tersoff = workload == "si_tersoff"
    kern, roofline, b_step = kernel_report(st, model.info, n, st_all, tersoff=tersoff,
                                           fused_angular=("partial_forces_in_one_kernel" in eng.describe() or "one_kernel_per_brick" in eng.describe()),
                                               brick_force="one_kernel_per_brick" in eng.describe())
    if tersoff and roofline:
        roofline["kernel"] = {"radial_descriptor": "tersoff_bond_order", "force_assemble": "tersoff_force"}[roofline["kernel"]]
    return {"workload": label, "metric": METRIC[workload], "value": n * steps / elapsed, "unit": "atom-steps/s",
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "dtype": "f64" if tersoff else "f32",
            "rebuilds_in_timed_region": int(st.num_rebuild - reb0), "kernel_forms": eng.describe(),
            "roofline": roofline, "step_algorithmic_bytes_per_atom": b_step,
            "step_hbm_frac": b_step * (n * steps / elapsed) / (HBM_PEAK_GBS * 1e9),
            "kernels_avg_ms": {k: round(v["avg_ms"], 5) for k, v in kern.items()}}


def run_decomposed(args, world, rank, dev, model, label, h_block, typ, x, mass, vel, scaling=None, leg=0, steps=None,
                   warmup=None, overlap=None, ghosts=None, checksum=False, brief=False):
    """N > 1 (or --decomposed on one GPU): the C++ domain-decomposed driver of libnepmi (nepmi_dist_*), one rank per
    GPU, ghost positions over RCCL/xGMI.  Python only builds the synthetic block and passes pointers.
    -> the bench line as a dict on rank 0 (None elsewhere).  `leg` numbers the runs of one process (own TCP port each)."""
    global STAGE
    scaling = scaling or args.scaling
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    import torch
    import torch.distributed as dist
    import gpumd_amd
    from gpumd_amd import structures as H
    from gpumd_amd.dist import DistMD, Transport, choose_grid

    lib = gpumd_amd.load_library()
    # Weak scaling stacks the ranks' blocks along the first lattice vector (N x 1 x 1 slabs, the reference's own
    # one-directional partition, force.cu:122-160): every rank has two neighbours and ghosts across two faces only -- a
    # 1,024,000-atom PbTe block gets +12 % local atoms (+6 % that need descriptors) instead of +36 % (+17 %) in a 2 x 2 x 2
    # arrangement of the same blocks.  Strong scaling cuts ONE system: there the most cubic grid has the least surface.
    grid = (world, 1, 1) if scaling == "weak" else choose_grid(world)
    Hb = np.asarray(h_block).reshape(3, 3)
    n = len(typ)
    STAGE = "decomposed[%s]: transport" % scaling
    if scaling == "strong":
        # the SAME global system on every N: rank r contributes a slice of it (setup migrates the atoms to their owners)
        Hg = Hb
        mine = np.arange(n) % world == rank
        X = np.ascontiguousarray(x.reshape(3, n)[:, mine])
        V = np.ascontiguousarray(vel.reshape(3, n)[:, mine])
        T, M = typ[mine], mass[mine]
    else:
        # weak scaling: every rank generates its own block of the global crystal (grid x block)
        Hg = Hb * np.asarray(grid, dtype=np.float64)[None, :]
        coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        X = x.reshape(3, n) + (Hb @ np.asarray(coords, dtype=np.float64))[:, None]
        V, T, M = vel.reshape(3, n), typ, mass
    global _TRANSPORT
    if _TRANSPORT is not None:  # one communicator for all the legs of a process (the first leg opened it)
        tr, transport = _TRANSPORT
    elif world > 1 and os.environ.get("NEPMI_DIST_BACKEND", "nccl") == "nccl":
        def bcast(ident):
            t = torch.zeros(128, dtype=torch.uint8, device=dev)
            if ident is not None:
                t.copy_(torch.frombuffer(bytearray(ident), dtype=torch.uint8))
            dist.broadcast(t, src=0)
            return bytes(t.cpu().numpy().tobytes())
        tr = Transport.rccl(lib, rank, world, bcast)
        transport = "RCCL send/recv of ghost positions over xGMI, skin vote all-reduced on the device"
    else:
        # functional check on a box with fewer GPUs than ranks (or one rank): host sockets, never a performance run
        tr = Transport.tcp(lib, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                           int(os.environ.get("MASTER_PORT", "29400")) + 1 + leg, rank, world)
        transport = "TCP sockets (host staging)" if world > 1 else "single rank"
    _TRANSPORT = (tr, transport)
    STAGE = "decomposed[%s]: setup (first decomposition, ghost exchange, list build)" % scaling
    if os.environ.get("NEPMI_BENCH_FAIL_STAGE") == "setup" and rank == world - 1:
        raise RuntimeError("injected failure (NEPMI_BENCH_FAIL_STAGE: tests/test_bench_launch.py)")
    ghosts = args.ghosts if ghosts is None else ghosts
    overlap = args.overlap if overlap is None else overlap
    md = DistMD(model, tr, Hg.reshape(9), (1, 1, 1), grid, ghost_mode=ghosts)
    if overlap >= 0:
        md.set_overlap(bool(overlap))
    md.setup(torch.from_numpy(np.ascontiguousarray(T)).to(dev), torch.from_numpy(np.ascontiguousarray(M)).to(dev),
             torch.from_numpy(np.ascontiguousarray(X).reshape(-1)).to(dev),
             torch.from_numpy(np.ascontiguousarray(V).reshape(-1)).to(dev))
    dt = 1.0 / H.TIME_UNIT
    ens = args.ensemble
    t_args = (300.0, 300.0, 100.0)
    STAGE = "decomposed[%s]: initial force (first ghost exchange of a step)" % scaling
    md.compute()
    STAGE = "decomposed[%s]: warm-up steps" % scaling
    if warmup > 0:
        md.run(ens, dt, warmup, *t_args)
    md.engine_set_timing(2)  # the dominant kernel only inside the timed region (see the single-GPU path)
    tr.rccl_stats(time_every=4, reset=True)  # (RCCL transports: count what the timed region moves, time every 4th exchange)
    dec0 = md.info().num_decompositions
    STAGE = "decomposed[%s]: timed region" % scaling
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    md.run(ens, dt, steps, *t_args)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rccl = tr.rccl_stats(time_every=0)
    STAGE = "decomposed[%s]: instrumented pass" % scaling
    st = md.engine_stats(with_lists=True)
    th = md.thermo()
    md.engine_set_timing(1)
    md.run(ens, dt, max(10, min(steps, 40)), *t_args)  # instrumented pass for the per-kernel table, outside the clock
    st_all = md.engine_stats(with_lists=False)
    md.engine_set_timing(0)
    info = md.info()
    # A checksum of the decomposed forces: every atom's force gathered on rank 0 and compared with ONE-domain evaluation of the
    # same positions there -- a transport that silently delivered wrong ghosts would produce a fast wrong number otherwise
    check = None
    if checksum:
        STAGE = "decomposed[%s]: checksum (gather_global + one-domain evaluation on rank 0)" % scaling
        nt = int(info.n_total)
        g_x = torch.zeros(3 * nt if rank == 0 else 1, dtype=torch.float64, device=dev)
        g_f = torch.zeros(3 * nt if rank == 0 else 1, dtype=torch.float64, device=dev)
        md.compute()  # forces of the current positions (the run loops leave the last step's kick pending)
        md.gather_global(0, pos=g_x, force=g_f)
        if rank == 0:
            try:
                # ids of the strong leg: rank r contributed atoms r, r + world, ... in that order: global id = position in the
                # concatenation of the ranks' slices
                ids_of = np.concatenate([np.arange(n)[np.arange(n) % world == r] for r in range(world)]) if scaling == "strong" else np.arange(nt)
                t_typ = torch.from_numpy(np.ascontiguousarray(typ[ids_of] if scaling == "strong" else np.tile(typ, world))).to(dev)
                one = gpumd_amd.NEP(model, nt)
                o_pe, o_f, o_w = (torch.zeros(k * nt, dtype=torch.float64, device=dev) for k in (1, 3, 9))
                one.force_compute(Hg.reshape(9), t_typ, g_x, o_pe, o_f, o_w)
                torch.cuda.synchronize()
                check = {"atoms": nt, "max_abs_dF_eV_per_A": float((o_f - g_f).abs().max().item()),
                         "max_abs_F": float(o_f.abs().max().item()),
                         "note": "decomposed forces (gather_global) vs one-domain nepmi_force_compute of the same positions on rank 0"}
                del one
            except Exception as e:  # never lose the line to the check
                check = {"error": "%s: %s" % (type(e).__name__, e)}
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_loc = torch.tensor([float(info.n_local)], dtype=torch.float64, device=dev)
    t_own = torch.zeros(world, dtype=torch.float64, device=dev)
    t_own[rank] = own
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_loc, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_own, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    out = None
    if rank == 0:
        forms = md.engine_describe()
        kern, roofline, b_step = kernel_report(st, model.info, info.n_local, st_all,
                                               fused_angular=("partial_forces_in_one_kernel" in forms or "one_kernel_per_brick" in forms),
                                               brick_force="one_kernel_per_brick" in forms)
        total = info.n_total
        per_rank_ms = [float(v) / steps * 1e3 for v in t_own.cpu().numpy()]
        out = {
            "metric": METRIC[args.workload] + (", " + ens if ens != "nve" else ""),
            "value": total * steps / elapsed, "unit": "atom-steps/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "dtype_note": "FP32 kernel arithmetic like the reference NEP path; FP64 positions, velocities and accumulated per-atom outputs",
            "data": "synthetic",
            "config": {"workload": label + (" (per GPU)" if scaling == "weak" else " (whole system, strong scaling)"),
                       "ensemble": ens, "atoms_total": total,
                       "parallelism": "spatial decomposition %dx%dx%d, %s, %s" % (grid + (
                           "ghost shell rc+skin, partial forces returned to the owners (two exchanges per step)"
                           if info.reverse_ghosts else "ghost shell 2(rc+skin), inner ring recomputed (one exchange per step)",
                           transport)),
                       "ghost_mode": "reverse" if info.reverse_ghosts else "forward",
                       "local_atoms_max": int(n_loc.item()), "decompositions_in_timed_region": int(info.num_decompositions - dec0),
                       "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms),
                                                "all": [round(v, 4) for v in per_rank_ms],
                                                "note": "each rank's own wall time of the timed region before the closing barrier"},
                       "steps_with_overlapped_exchange": int(info.num_overlapped),
                       "mean_nn_radial": st.mean_nn_radial, "mean_nn_angular": st.mean_nn_angular},
            "roofline": roofline, "step_algorithmic_bytes_per_atom": b_step,
            "step_hbm_frac": b_step * (total * steps / elapsed) / (HBM_PEAK_GBS * 1e9 * world),
            "kernels": kern, "thermo_last": [float(v) for v in th],
        }
        free_b, total_b = torch.cuda.mem_get_info(dev)
        out["device_memory"] = {"used_gb_rank0": (total_b - free_b) / 1e9, "total_gb": total_b / 1e9,
                                "note": "hipMemGetInfo after the run on rank 0: engine + decomposition buffers + the PyTorch context"}
        out["config"]["kernel_forms"] = md.engine_describe()
        out["config"]["overlap"] = int(overlap)
        if rccl is not None:
            ex = max(int(rccl["exchanges"]), 1)
            out["rccl"] = {"comm_nranks": int(rccl["comm_nranks"]), "comm_rank": int(rccl["comm_rank"]),
                           "exchanges_in_timed_region": int(rccl["exchanges"]), "exchanges_per_step": rccl["exchanges"] / steps,
                           "messages_per_exchange": rccl["messages"] / ex, "bytes_sent_per_exchange": rccl["bytes_sent"] / ex,
                           "allreduces_per_step": rccl["allreduces"] / steps,
                           "us_per_exchange": rccl["us_per_timed_exchange"], "timed_exchanges": int(rccl["timed_exchanges"]),
                           "note": "rank 0's RCCL transport over the timed region; us_per_exchange: HIP events on the stream of every "
                                   "4th grouped ncclSend/ncclRecv exchange (it includes waiting for the slowest peer)"}
        if check is not None:
            out["checksum"] = check
        if brief:
            out = {k: out[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling", "rccl", "checksum") if k in out}
            out.update(overlap=int(overlap), ghost_mode="reverse" if info.reverse_ghosts else "forward",
                       local_atoms_max=int(n_loc.item()))
    STAGE = "decomposed[%s]: teardown" % scaling
    md.close()
    if world > 1:
        dist.barrier()
    return out


STAGE = "start"  # what the process was doing when it failed (the "error" line names it)
_TRANSPORT = None  # (gpumd_amd.dist.Transport, description): opened by the first decomposed leg, closed when the process ends


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run, one rank per GPU
    (the reference starts its multi-GPU path from one process, src/force/force.cu:122-160; here one process drives one
    GPU, so the plain command spawns them).  On a box with fewer GPUs than ranks the ranks share the devices over the
    TCP transport: a functional run of the same protocol, labelled as such in the line."""
    import socket
    import subprocess
    import torch
    env = dict(os.environ)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev == 0:
        print(json.dumps({"error": "no GPU visible", "stage": "self-launch", "n_gpus": args.gpus}))
        raise SystemExit(2)
    if ndev < args.gpus and "NEPMI_DIST_BACKEND" not in env:
        env["NEPMI_DIST_BACKEND"] = "tcp"
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the children's stdout is filtered down to ONE JSON line: rank 0's bench line, or -- when the launch failed -- one
    # {"error": ...} line that names the stage of the first rank that reported one (failed ranks print theirs on either stream)
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    import threading
    errs, lines = [], []

    def pump(src, sink, keep):
        for ln in src:
            if ln.startswith("{") and ('"error"' in ln or '"metric"' in ln):
                keep.append(ln.strip())
            else:
                sink.write(ln)
                sink.flush()
    th = [threading.Thread(target=pump, args=(pr.stdout, sys.stderr, lines)),
          threading.Thread(target=pump, args=(pr.stderr, sys.stderr, errs))]
    for t in th:
        t.start()
    rc = pr.wait()
    for t in th:
        t.join()
    good = [ln for ln in lines if '"error"' not in ln[:20] and '"value"' in ln]
    if good:
        print(good[0])
    else:
        bad = [ln for ln in lines + errs if ln.startswith('{"error"')]
        if bad:
            print(bad[0])
        else:
            print(json.dumps({"error": "the %d-rank launch exited with code %d before any rank reported" % (args.gpus, rc),
                              "stage": "torch.distributed.run", "n_gpus": args.gpus}))
        rc = rc or 3
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, nargs=3, default=[16, 16, 16], help="replicate na nb nc of the 250-atom cell")
    ap.add_argument("--workload", default="pbte", choices=["pbte", "pbte_ortho", "carbon", "carbon2024", "unep", "si_tersoff"],
                    help="pbte = BASELINE config 3 (the bench line); the others are extra single-GPU measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip extra_measurements (config 2, the rebuild-inclusive PbTe segment, the config 4/5 model families; "
                         "N > 1: the strong-scaling leg), all timed after the bench line's clock")
    ap.add_argument("--decomposed", action="store_true",
                    help="run the N > 1 code path (the C++ domain-decomposed driver) even on one GPU, to measure its overhead")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU gets its own block of --reps cells (default); strong = the --reps system is "
                         "shared by all GPUs (SURVEY.md 8e: >= 6x at 8 GPUs on the 1 M-atom config)")
    ap.add_argument("--ghosts", type=int, default=-1, choices=[-1, 0, 1],
                    help="decomposed runs, nepmi_dist_set_ghost_mode: -1 the counted rule (default), 0 forward, 1 reverse ghosts")
    ap.add_argument("--overlap", type=int, default=-1, choices=[-1, 0, 1],
                    help="decomposed runs, nepmi_dist_set_overlap: exchange on the side stream while interior bricks run "
                         "(-1: library default; 0 / 1 for an A/B in one command)")
    ap.add_argument("--ensemble", default="nve", choices=["nve", "nvt_ber", "nvt_nhc", "nvt_bdp", "nvt_lan", "nvt_bao"],
                    help="decomposed runs: the ensemble (config 5 is NVT); the single-GPU bench line is NVE")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:  # one instance of the all-cores CPU aggregate, no GPU involved
        cpu_worker(args.cpu_worker)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    try:
        bench(args)
    except SystemExit:
        raise
    except BaseException as e:  # a failed rank says where it failed instead of dying silently (rank 0's is the bench line)
        line = json.dumps({"error": "%s: %s" % (type(e).__name__, e), "stage": STAGE, "rank": rank, "n_gpus": args.gpus,
                           "metric": METRIC.get(args.workload, args.workload)})
        (sys.stdout if rank == 0 else sys.stderr).write(line + "\n")
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)  # do not wait in a collective's destructor for ranks that are gone


def bench(args):
    global STAGE
    import torch
    import torch.distributed as dist
    import gpumd_amd
    from gpumd_amd import structures as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    # NEPMI_DIST_BACKEND=tcp|gloo: functional check of the N > 1 path on a box with fewer GPUs than
    # ranks (ranks share devices, messages staged through host memory); never a performance run.
    backend = os.environ.get("NEPMI_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        STAGE = "torch.distributed init (%s)" % backend
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # a wedged exchange should fail fast (watchdog), not hold the node for the default 10 min
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))

    # ---- workload: every rank generates its own block of `reps` cells (weak scaling) ----
    STAGE = "workload"
    reps = tuple(args.reps)
    label, nep_txt, h, typ, x, mass, vel = build_workload(args.workload, reps, 42 + (rank if args.scaling == "weak" else 0))
    n = len(typ)
    x0, vel0 = x.copy(), vel.copy()
    model = gpumd_amd.Model(nep_txt)
    dt = 1.0 / H.TIME_UNIT
    if world > 1 or args.decomposed:
        # N > 1: BEFORE the clock, a short probe of exchange/compute overlap {off, on} x ghosts {forward, reverse} (8 steps each on
        # the headline's own system); the headline runs the combination that won and says so in `config` -- the defaults rest on a
        # one-GPU proxy where a collective is an event wait, and the line that gets recorded should be the best form the code has.
        # Flags given on the command line are kept.  Every rank takes the same decision (rank 0's, broadcast).
        probe = None
        pick_ov, pick_gh = None, None
        if world > 1 and args.overlap < 0 and args.ghosts < 0 and not args.no_extras:
            probe = {}
            leg = 100
            for ov in (0, 1):
                for gh in (0, 1):
                    key = "overlap%d_%s" % (ov, "reverse" if gh else "forward")
                    try:
                        r = run_decomposed(args, world, rank, dev, model, label, h, typ, x.copy(), mass, vel.copy(), leg=leg, steps=8, warmup=3,
                                           overlap=ov, ghosts=gh, brief=True)
                        if r is not None:
                            probe[key] = {"ms_per_step": r["ms_per_step"], "overlap": ov, "ghosts": gh}
                    except Exception as e:
                        probe[key] = {"error": "%s: %s" % (type(e).__name__, e), "stage": STAGE}
                    leg += 1
            choice = torch.zeros(2, dtype=torch.int64, device=dev) - 1
            if rank == 0:
                ok = [v for v in probe.values() if "ms_per_step" in v]
                if ok:
                    best = min(ok, key=lambda v: v["ms_per_step"])
                    choice[0], choice[1] = best["overlap"], best["ghosts"]
            dist.broadcast(choice, src=0)
            if int(choice[0].item()) >= 0:
                pick_ov, pick_gh = int(choice[0].item()), int(choice[1].item())
        out = run_decomposed(args, world, rank, dev, model, label, h, typ, x, mass, vel, overlap=pick_ov, ghosts=pick_gh)
        if out is not None and probe is not None:
            out["config"]["form_probe_before_the_clock"] = {"results": probe, "chosen": {"overlap": pick_ov, "ghosts": "reverse" if pick_gh else "forward"},
                                                            "note": "8 steps of each combination on this system before the clock; the headline ran the fastest"}
        if world > 1 and backend != "nccl" and out is not None:
            out["functional_only"] = ("ranks share %d GPU(s) over the TCP transport (host staging): the protocol of the N-GPU run, "
                                      "not its performance" % torch.cuda.device_count())
        if world > 1 and args.scaling == "weak" and not args.no_extras:
            # after the clock: the SAME 1,024,000-atom system as the N = 1 line cut over the N GPUs (strong scaling), with the
            # force checksum against a one-domain evaluation on rank 0
            try:
                label_s, _, h_s, typ_s, x_s, mass_s, vel_s = build_workload(args.workload, reps, 42)
                strong = run_decomposed(args, world, rank, dev, model, label_s, h_s, typ_s, x_s, mass_s, vel_s,
                                        scaling="strong", leg=1, steps=min(args.steps, 100), warmup=min(args.warmup, 10), checksum=True)
                if out is not None and strong is not None:
                    out.setdefault("extra_measurements", {})["strong"] = {
                        k: strong[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "config", "roofline",
                                               "step_hbm_frac", "rccl", "checksum") if k in strong}
            except Exception as e:  # never lose the bench line to an extra
                if out is not None:
                    out.setdefault("extra_measurements", {})["strong"] = {"error": "%s: %s" % (type(e).__name__, e), "stage": STAGE}
            # ... and the 2 x 2 matrix exchange/compute overlap {off, on} x ghosts {forward, reverse} on both legs (40 steps
            # each), so that ONE driver command settles the defaults on real hardware
            matrix = {}
            leg = 2
            for sc in ("weak", "strong"):
                for ov in (0, 1):
                    for gh in (0, 1):
                        key = "%s_overlap%d_%s" % (sc, ov, "reverse" if gh else "forward")
                        try:
                            if sc == "weak":
                                wl = (label, h, typ, x, mass, vel)
                            else:
                                wl = (label_s, h_s, typ_s, x_s, mass_s, vel_s)
                            r = run_decomposed(args, world, rank, dev, model, wl[0], wl[1], wl[2], wl[3].copy(), wl[4], wl[5].copy(),
                                               scaling=sc, leg=leg, steps=30, warmup=5, overlap=ov, ghosts=gh, brief=True)
                            if r is not None:
                                matrix[key] = r
                        except Exception as e:
                            matrix[key] = {"error": "%s: %s" % (type(e).__name__, e), "stage": STAGE}
                        leg += 1
            if out is not None:
                out.setdefault("extra_measurements", {})["overlap_x_ghosts"] = matrix
        if out is not None:
            print(json.dumps(out))
            sys.stdout.flush()
        if _TRANSPORT is not None:
            _TRANSPORT[0].close()
        if world > 1:
            dist.destroy_process_group()
        return
    eng = gpumd_amd.NEP(model, n)
    t_type = torch.from_numpy(typ).to(dev)
    t_mass = torch.from_numpy(mass).to(dev)
    t_x = torch.from_numpy(x).to(dev)
    t_v = torch.from_numpy(vel).to(dev)
    t_pe = torch.zeros(n, dtype=torch.float64, device=dev)
    t_f = torch.zeros(3 * n, dtype=torch.float64, device=dev)
    t_w = torch.zeros(9 * n, dtype=torch.float64, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # kernel-variant switches for A/B measurements (defaults = what the engine picks itself)
    if "NEPMI_BENCH_TILES" in os.environ:
        eng.set_tiles(int(os.environ["NEPMI_BENCH_TILES"]))
    if "NEPMI_BENCH_RECOMPUTE" in os.environ:
        eng.set_angular_recompute(int(os.environ["NEPMI_BENCH_RECOMPUTE"]))
    if "NEPMI_BENCH_MFMA" in os.environ:
        eng.set_mfma(os.environ["NEPMI_BENCH_MFMA"] != "0")
    if "NEPMI_BENCH_LANES" in os.environ:
        eng.set_win_lanes(int(os.environ["NEPMI_BENCH_LANES"]))
    if "NEPMI_BENCH_ANGFUSED" in os.environ:  # angular descriptor + ANN + partial forces: 1 one kernel (default), 0 separate
        eng.set_angular_fused(os.environ["NEPMI_BENCH_ANGFUSED"] != "0")
    if "NEPMI_BENCH_BRICK" in os.environ:  # one force kernel per brick (default) / fused angular kernel + scatter kernel
        eng.set_brick_force(os.environ["NEPMI_BENCH_BRICK"] != "0")
    if "NEPMI_BENCH_SYNC" in os.environ:  # per-step radial list of the scatter-form steps: 1 wave-synchronous words (default), 0 slot-major compact list
        eng.set_radial_sync(os.environ["NEPMI_BENCH_SYNC"] != "0")
    if "NEPMI_BENCH_FORM" in os.environ:  # force assembly: 0 gather, 1 scatter (default: the run loops' rule)
        eng.set_force_form(int(os.environ["NEPMI_BENCH_FORM"]))
    # initial force (Run::perform_a_run computes it before the loop), then warm-up steps
    eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w)
    if args.warmup > 0:
        eng.run_nve(h, t_type, t_mass, dt, args.warmup, t_x, t_v, t_pe, t_f, t_w)
    # Inside the timed region only the dominant kernel (force assembly) carries HIP events -- two records per step; a
    # record around EVERY kernel stops them from running back to back and costs about 5 % of the step.  The full
    # per-kernel table comes from an instrumented pass of further steps after the clock has stopped.
    dom_slot = dominant_slot(eng, lambda k: eng.run_nve(h, t_type, t_mass, dt, k, t_x, t_v, t_pe, t_f, t_w, thermo_every=k))
    eng.set_timing(16 + dom_slot)
    reb0 = eng.stats().num_rebuild
    barrier()
    t0 = time.perf_counter()
    th = eng.run_nve(h, t_type, t_mass, dt, args.steps, t_x, t_v, t_pe, t_f, t_w, thermo_every=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    st = eng.stats(with_lists=True)
    eng.set_timing(1)
    extra = max(10, min(args.steps, 60))
    eng.run_nve(h, t_type, t_mass, dt, extra, t_x, t_v, t_pe, t_f, t_w, thermo_every=extra)
    st_all = eng.stats(with_lists=False)
    eng.set_timing(0)

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())
    total_atoms = n * world
    value = total_atoms * args.steps / elapsed

    if rank == 0:
        tersoff = args.workload == "si_tersoff"
        kern, roofline, b_step = kernel_report(st, model.info, n, st_all, tersoff=tersoff,
                                               fused_angular=("partial_forces_in_one_kernel" in eng.describe() or "one_kernel_per_brick" in eng.describe()),
                                               brick_force="one_kernel_per_brick" in eng.describe())
        if tersoff:
            # the two Tersoff kernels sit in the radial and force-assembly slots of the engine's timing table
            kern = {{"radial_descriptor": "tersoff_bond_order", "force_assemble": "tersoff_force"}.get(k, k): v
                    for k, v in kern.items()}
            if roofline:
                roofline["kernel"] = {"radial_descriptor": "tersoff_bond_order", "force_assemble": "tersoff_force"}[roofline["kernel"]]
                roofline["note"] = ("13,824 atoms = 54 atoms per CU: the step is launch-latency bound (7 kernels of 3-8 us "
                                    "each), far from any throughput roofline; FP64 arithmetic like the reference's Tersoff")
        out = {
            "metric": METRIC[args.workload],
            "value": value, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if tersoff else "f32",
            "dtype_note": ("FP64 throughout like the reference's Tersoff kernels" if tersoff else
                           "FP32 kernel arithmetic like the reference NEP path; FP64 positions, velocities and accumulated per-atom outputs"),
            "data": "synthetic",
            "config": {"workload": label,
                       "atoms_total": total_atoms, "rebuilds_in_timed_region": int(st.num_rebuild - reb0),
                       "mean_nn_radial": st.mean_nn_radial, "mean_nn_angular": st.mean_nn_angular,
                       "lds_window_mode": int(st.radial_tiles), "parallelism": "1 GPU",
                       "kernel_forms": eng.describe(),
                       "ann_form": ("per-atom descriptor + ANN fused in one kernel, packed FP32 on the vector units (no MFMA in this "
                                    "workload's step)" if "ann=fused" in eng.describe() else
                                    ("matrix-core ANN kernel (v_mfma_f32_32x32x2_f32)" if "ann=mfma" in eng.describe() else "per-atom ANN kernel")),
                       "thermo_every": args.steps,
                       "thermo_note": "thermo reduced once, at the last step of the timed region (the reference's Ensemble reduces it "
                                      "every step, ensemble_nve.cu:59-95: a legitimate saving of the fused loop, stated here)"},
            "roofline": roofline,
            "step_algorithmic_bytes_per_atom": b_step,
            "step_hbm_frac": b_step * (n * args.steps / elapsed) / (HBM_PEAK_GBS * 1e9),
            "kernels": kern,
            "kernels_note": "the step's longest force kernel (found by a 4-step instrumented probe before the clock): HIP events inside "
                            "the timed region; the others: an instrumented pass of %d further steps" % extra,
            "gpu_state": gpu_state(),
            "thermo_last": [float(v) for v in th[-1]] if len(th) else None,
        }
        if args.workload == "pbte":
            # the path is FP32-VALU/gather bound, so the step is also priced against the
            # FP32 vector peak with the survey's FLOP count of the reference algorithm (8e4 per atom-step)
            fl = own_flops(model.info, st.max_nn_skin and (st.mean_nn_radial * 1.31), st.mean_nn_radial, st.mean_nn_angular)
            out["fp32_valu"] = {"flop_per_atom_step_own": fl["total"], "frac_own": value * fl["total"] / 157.3e12,
                                "own_flop_by_kernel": {k: round(v) for k, v in fl.items() if k != "total"},
                                "peak_tflops": 157.3,
                                "flop_per_atom_step_survey": 8.0e4, "frac_survey_equivalent": value * 8.0e4 / 157.3e12,
                                "note": "frac_own prices the operations this engine executes (counted from its kernels for "
                                        "this model shape; ~1.31 Verlet candidates per radial neighbour); the survey figure is "
                                        "the reference algorithm's count and only says how much of it was avoided"}
        if world == 1 and not args.no_extras and not args.no_cpu_baseline and args.workload == "pbte":
            # After the clock: (a) the same engine for further 100-step segments until one of them contains a list rebuild
            # (at 300 K one rebuild per ~100 steps): the rebuild-inclusive rate; (b) BASELINE config 2 (Si 13,824 atoms,
            # Tersoff-1989, FP64) on its own engine.  Each carries its own roofline object.
            extras = {}
            try:
                seg = None
                for _ in range(4):
                    r0 = eng.stats().num_rebuild
                    eng.set_timing(16 + dom_slot)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    eng.run_nve(h, t_type, t_mass, dt, 100, t_x, t_v, t_pe, t_f, t_w, thermo_every=100)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                    st2 = eng.stats(with_lists=True)
                    eng.set_timing(0)
                    kern2, roof2, _ = kernel_report(st2, model.info, n, None,
                                                    fused_angular=("partial_forces_in_one_kernel" in eng.describe() or "one_kernel_per_brick" in eng.describe()),
                                               brick_force="one_kernel_per_brick" in eng.describe())
                    seg = {"workload": label, "steps": 100, "ms_per_step": el / 100 * 1e3, "value": n * 100 / el,
                           "unit": "atom-steps/s", "rebuilds_in_timed_region": int(st2.num_rebuild - r0), "roofline": roof2}
                    if seg["rebuilds_in_timed_region"] >= 1:
                        break
                extras["pbte_100_steps_with_rebuild"] = seg
            except Exception as e:  # never lose the bench line to an extra
                extras["pbte_100_steps_with_rebuild"] = {"error": str(e)}
            # (a2) like for like with Ensemble_NVE::compute2 (ensemble_nve.cu:59-95 reduces thermo every step): thermo_every = 1,
            #      i.e. energies and virials written and reduced on every step
            try:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.run_nve(h, t_type, t_mass, dt, 40, t_x, t_v, t_pe, t_f, t_w, thermo_every=1)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                extras["pbte_thermo_every_step"] = {"workload": label, "steps": 40, "thermo_every": 1, "ms_per_step": el / 40 * 1e3,
                                                    "value": n * 40 / el, "unit": "atom-steps/s"}
            except Exception as e:
                extras["pbte_thermo_every_step"] = {"error": str(e)}
            # (a3) the drop-in entry as a GPUMD maintainer would call it (force.cu:819-822 inside
            #      run.cu:250-326): per step nepmi_vv_step1 -> nepmi_force_compute -> nepmi_vv_step2 -> nepmi_find_thermo on the
            #      caller's arrays -- gather-form assembly, all thirteen outputs, scatter-add into caller order, every step
            try:
                vol = float(abs(np.linalg.det(np.asarray(h, dtype=np.float64).reshape(3, 3))))
                t_th = torch.zeros(8, dtype=torch.float64, device=dev)
                ksteps = 30
                for timed_pass in (False, True):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(ksteps if timed_pass else 3):
                        eng.vv_step1(dt, t_mass, t_f, t_x, t_v)
                        eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w)
                        eng.vv_step2(dt, t_mass, t_f, t_v)
                        eng.find_thermo(vol, t_mass, t_pe, t_v, t_w, t_th)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                extras["pbte_per_call_dropin"] = {"workload": label, "steps": ksteps, "ms_per_step": el / ksteps * 1e3,
                                                  "value": n * ksteps / el, "unit": "atom-steps/s",
                                                  "kernel_forms": eng.describe(),
                                                  "note": "one host call per stage and step through the C ABI (what gpumd_ref_mi executes); "
                                                          "thermo every step"}
                # ... and with nepmi_engine_set_virial_mode(e, 1): a host that needs the TOTAL virial only (no heat current, no
                # per-atom virial dump) lets the per-call evaluations take the scatter form as well
                eng.set_virial_mode(1)
                for timed_pass in (False, True):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(ksteps if timed_pass else 3):
                        eng.vv_step1(dt, t_mass, t_f, t_x, t_v)
                        eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w)
                        eng.vv_step2(dt, t_mass, t_f, t_v)
                        eng.find_thermo(vol, t_mass, t_pe, t_v, t_w, t_th)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                extras["pbte_per_call_dropin_totals"] = {"workload": label, "steps": ksteps, "ms_per_step": el / ksteps * 1e3,
                                                         "value": n * ksteps / el, "unit": "atom-steps/s", "kernel_forms": eng.describe(),
                                                         "note": "the same sequence with nepmi_engine_set_virial_mode(e, 1)"}
                eng.set_virial_mode(0)
            except Exception as e:
                extras["pbte_per_call_dropin"] = {"error": str(e)}
            # (a4) the run-time-shape kernels (any n_max / basis_size / l_max / neuron count: what a user-trained potential
            #      outside the compiled shapes gets), forced on this model so that the same workload is compared
            try:
                extras["pbte_generic_shape"] = measure_extra("pbte", reps, 10, 3, dev, generic=True)
            except Exception as e:
                extras["pbte_generic_shape"] = {"error": str(e)}
            # (a4') ... and what such a model gets WITHOUT a compiler: zero-padded into a compiled cover shape (nep_model.h:
            #      embed_model; forced here on the same PbTe model and on C_2024, whose own shapes have kernels / a JIT core)
            os.environ["NEPMI_FORCE_COVER"] = "1"
            try:
                for key, wl, rp, k in (("pbte_zero_padded_into_cover_shape", "pbte", reps, 20), ("c2024_512k_zero_padded_into_cover_shape", "carbon2024", (10, 10, 10), 5)):
                    try:
                        extras[key] = measure_extra(wl, rp, k, 3, dev)
                    except Exception as e:
                        extras[key] = {"error": str(e)}
            finally:
                os.environ.pop("NEPMI_FORCE_COVER", None)
            # (a5) a shipped potential of ANOTHER shape (C_2024: n_max 12 8, basis 16 12): kernels compiled for its shape (a JIT
            #      core, prebuilt by build(): NEPMI_JIT=2 = never the compiler here) against the run-time-shape kernels
            for key, mode in (("c2024_512k_jit_core", "2"), ("c2024_512k_run_time_shape", "0")):
                old_mode = os.environ.get("NEPMI_JIT")
                os.environ["NEPMI_JIT"] = mode
                try:
                    extras[key] = measure_extra("carbon2024", (10, 10, 10), 10 if mode == "2" else 3, 2, dev)
                    # (a prebuilt core carries the hash of the sources it was compiled from: a stale one is not loaded -- say so)
                    extras[key]["served_by"] = ("JIT core of the model's shape" if "shape=jit(" in extras[key].get("kernel_forms", "")
                                                else "run-time-shape kernels")
                except Exception as e:
                    extras[key] = {"error": str(e)}
                finally:
                    if old_mode is None:
                        os.environ.pop("NEPMI_JIT", None)
                    else:
                        os.environ["NEPMI_JIT"] = old_mode
            try:
                extras["config2_si_tersoff"] = measure_extra("si_tersoff", (16, 16, 16), 2000, 200, dev)
            except Exception as e:
                extras["config2_si_tersoff"] = {"error": str(e)}
            # (c) the model families of configs 4 and 5 at one million atoms on this GPU (what every GPU of those 8-GPU
            # configurations executes per million owned atoms), each on its own engine, same protocol
            for key, wl, rp in (("config4_model_unep_1m", "unep", (16, 16, 16)), ("config5_model_carbon_1m", "carbon", (10, 10, 10))):
                try:
                    extras[key] = measure_extra(wl, rp, 40, 10, dev)
                except Exception as e:
                    extras[key] = {"error": str(e)}
            # the three like-for-like figures of SURVEY 8(d)'s run loop, at the top level (in front of the long extras block)
            lfl = {}
            for key, src in (("rebuild_inclusive_100_steps", "pbte_100_steps_with_rebuild"), ("thermo_every_step", "pbte_thermo_every_step"),
                             ("per_call_dropin_sequence", "pbte_per_call_dropin"), ("per_call_dropin_total_virial_only", "pbte_per_call_dropin_totals")):
                e = extras.get(src) or {}
                if "value" in e:
                    lfl[key] = {"value": e["value"], "ms_per_step": e["ms_per_step"]}
                    if "rebuilds_in_timed_region" in e:
                        lfl[key]["rebuilds"] = e["rebuilds_in_timed_region"]
            lfl["note"] = ("the headline's timed region holds %d list rebuild(s) and one thermo reduction; SURVEY 8(d) defines the metric over a run loop "
                           "WITH rebuilds and thermo: these are the same engine on the same system, measured after the clock" % int(st.num_rebuild - reb0))
            out["like_for_like"] = lfl
            out["extra_measurements"] = extras
        if world == 1 and not args.no_cpu_baseline and args.workload == "pbte":
            with _stdout_to_stderr():
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
                del eng  # the reference needs the HBM of a 1 M-atom run of its own
                torch.cuda.empty_cache()
                ref_gpu = reference_gpu_baseline()
                if ref_gpu:
                    ref_gpu["speedup_of_this_engine"] = value / ref_gpu["value"]
                    out["cpu_baseline"]["reference_gpu_same_box"] = ref_gpu
        if world == 1 and not args.no_cpu_baseline and tersoff:
            with _stdout_to_stderr():
                out["cpu_baseline"] = cpu_baseline_tersoff(nep_txt, h, typ, x0, mass, vel0, args.cpu_seconds)
        # the last key of the line: what a reader who only sees the END of this (long) line needs
        out["tail_summary"] = {"value": value, "ms_per_step": out["ms_per_step"], "n_gpus": world, "workload": label,
                               "roofline_kernel": roofline and roofline["kernel"], "roofline_frac": roofline and roofline["frac"],
                               "step_hbm_frac": out["step_hbm_frac"], "like_for_like": out.get("like_for_like"),
                               "cpu_baseline_value": (out.get("cpu_baseline") or {}).get("value")}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
