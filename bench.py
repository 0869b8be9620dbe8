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
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
KERNEL_NAMES = ["gather_skin_check", "radial_descriptor", "angular_descriptor", "ann", "angular_partial_force",
                "force_assemble", "velocity_verlet", "list_rebuild"]


def build_pbte(reps, rattle=0.02, seed=42, temperature=300.0):
    import helpers as H
    fr = H.read_xyz_frames(H.golden("PbTe", "model.xyz"))[0]
    typ0 = H.types_from_species(fr["species"], ["Te", "Pb"])
    h, typ, pos = H.replicate(fr["h"], typ0, fr["pos"], reps)
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, rattle, pos.shape)
    typ = typ.astype(np.int32)
    mass = np.where(typ == 0, H.MASS["Te"], H.MASS["Pb"]).astype(np.float64)
    vel = H.maxwell_velocities(mass, temperature, seed=seed + 1)
    return h, typ, H.soa(pos), mass, vel


def build_workload(name, reps, seed):
    """-> (label, nep.txt, h, type, x_soa, mass, vel).  `pbte` is BASELINE config 3 (the bench line);
    `carbon` (C_2022_NEP4, diamond) and `unep` (UNEP-v1, 16-metal fcc alloy) are the model families of
    configs 5 and 4, offered for single-GPU kernel measurements."""
    import helpers as H
    if name == "pbte":
        h, typ, x, mass, vel = build_pbte(reps, seed=seed)
        return ("PbTe %d atoms/GPU (replicate %d %d %d of the 250-atom cell), NEP NVE, dt 1 fs, 300 K"
                % ((len(typ),) + tuple(reps)), H.golden("PbTe", "nep.txt"), h, typ, x, mass, vel)
    if name == "carbon":
        cells = tuple(5 * r for r in reps)  # 16 16 16 -> 80^3 diamond cells = 4,096,000 atoms; use --reps 10 10 10 for 1 M
        h, typ, x = H.diamond(cells, 3.57, rattle=0.02, seed=seed)
        mass = np.full(len(typ), H.MASS["C"])
        return ("diamond C %d atoms/GPU (%dx%dx%d cells), C_2022_NEP4, NVE" % ((len(typ),) + cells),
                H.golden("C", "nep.txt"), h, typ, x, mass, H.maxwell_velocities(mass, 300.0, seed=seed + 1))
    if name == "unep":
        cells = tuple(4 * r for r in reps)
        h, typ, x = H.fcc_alloy(cells, 3.9, 16, rattle=0.02, seed=seed)
        mass = np.full(len(typ), 100.0)
        return ("fcc 16-metal alloy %d atoms/GPU (%dx%dx%d cells), UNEP-v1 + ZBL, NVE" % ((len(typ),) + cells),
                H.golden("UNEP", "nep.txt"), h, typ, x, mass, H.maxwell_velocities(mass, 300.0, seed=seed + 1))
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
    import helpers as H
    nep = H.golden("PbTe", "nep.txt")
    h, typ, x, mass, vel = build_pbte(reps)
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
    import helpers as H
    n, calls, el, kind = _cpu_loop((4, 4, 4), seconds, 50)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)) if kind == "reference" else 1
    out = {"value": n * calls / el, "unit": "atom-steps/s", "cores": cores, "kind": kind,
           "sample": "PbTe %d atoms (replicate 4 4 4), %d NVE steps, %.1f s; NEP_CPU is serial outside its descriptor loop"
                     % (n, calls, el)}
    agg = cpu_aggregate(max(4.0, 0.75 * seconds))
    if agg:
        out["all_cores_aggregate"] = agg
    return out


def kernel_report(st, info, n_atoms_per_launch, st_all=None):
    """per-kernel mean durations (HIP events) + the roofline object of the dominant force kernel.
    st: stats of the timed region (timing mode 2: only the force-assembly slot is filled); st_all: stats of the
    instrumented pass after it (every slot) -- the timed region's own figure wins where both exist."""
    per_kernel, b_step = algorithmic_bytes(info, st.mean_nn_radial, st.mean_nn_angular)
    kern = {}
    for src in (st_all, st):
        if src is None:
            continue
        for k, name in enumerate(KERNEL_NAMES):
            if src.launches[k] > 0 and src.ms_kernel_sum[k] > 0.0:
                # launches of speculatively enqueued steps that were re-run after a list rebuild returned at once:
                # they are not work, the mean is over the launches that ran
                ran = int(src.launches[k]) - (int(src.discarded_steps) if 1 <= k <= 6 else 0)
                ran = max(ran, 1)
                kern[name] = {"launches": ran, "avg_ms": src.ms_kernel_sum[k] / ran}
    force_kernels = [k for k in kern if k in per_kernel and k not in ("velocity_verlet", "gather_skin_check")]
    dom = max(force_kernels, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"]) if force_kernels else None
    roofline = None
    if dom:
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            # HBM-side bytes per launch from a separate rocprofv3 PMC pass of this same command
            # (profiles/*_pmc_*.csv); only meaningful for the workload size it was taken on
            tj = json.load(open(tfile))
            if tj.get("atoms") == n_atoms_per_launch and dom in tj.get("kernels", {}):
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
        achieved = per_kernel[dom] * n_atoms_per_launch / (kern[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": per_kernel[dom] * n_atoms_per_launch,
                    "algorithmic_bytes_per_atom": per_kernel[dom], "avg_launch_ms": kern[dom]["avg_ms"],
                    "note": "FP32-VALU/gather bound stage (SURVEY.md 8d); see step_hbm_frac for the whole step"}
    return kern, roofline, b_step


def run_decomposed(args, world, rank, dev, model, h_block, typ, x, mass, vel, reps):
    """N > 1: spatial decomposition (gpumd_amd/domain.py), one rank per GPU, ghost positions over RCCL."""
    import torch
    import torch.distributed as dist
    import gpumd_amd
    import helpers as H
    from gpumd_amd.domain import DomainMD, choose_grid

    grid = choose_grid(world)
    Hb = np.asarray(h_block).reshape(3, 3)
    Hg = Hb * np.asarray(grid, dtype=np.float64)[None, :]       # global cell = grid x block
    coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
    n = len(typ)
    offset = Hb @ np.asarray(coords, dtype=np.float64)            # this rank's block of the global crystal
    X = torch.from_numpy(x.reshape(3, n) + offset[:, None]).to(dev)
    V = torch.from_numpy(vel.reshape(3, n).copy()).to(dev)
    T = torch.from_numpy(typ).to(dev)
    M = torch.from_numpy(mass).to(dev)
    staged = os.environ.get("NEPMI_DIST_BACKEND", "nccl") != "nccl"
    md = DomainMD(lambda cap: gpumd_amd.NEP(model, cap), model.info.rc_radial, Hg.reshape(9), (1, 1, 1), grid, rank,
                  world, dev, stage_through_host=staged)
    md.setup(X, V, T, M)
    dt = 1.0 / H.TIME_UNIT
    md.initial_forces()
    md.run(args.warmup, dt)
    md.engine.set_timing(2)  # the dominant kernel only inside the timed region (see the single-GPU path)
    dec0 = md.num_decompositions
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    md.run(args.steps, dt)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = md.engine.stats(with_lists=True)
    th = md.thermo()
    md.engine.set_timing(1)
    md.run(max(10, min(args.steps, 40)), dt)  # instrumented pass for the per-kernel table, outside the clock
    st_all = md.engine.stats(with_lists=False)
    md.engine.set_timing(0)
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_loc = torch.tensor([md.n_loc], dtype=torch.float64, device=dev)
    if world > 1:
        md._all_reduce(t_el, dist.ReduceOp.MAX)
        md._all_reduce(n_loc, dist.ReduceOp.MAX)
    elapsed = float(t_el.item())
    if rank == 0:
        kern, roofline, b_step = kernel_report(st, model.info, md.n_loc, st_all)
        total = md.n_total
        out = {
            "metric": "atom-steps/sec, NEP PbTe NVE (nep4 2 Te Pb, examples/nep_train/nep.txt)",
            "value": total * args.steps / elapsed, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_note": "FP32 kernel arithmetic like the reference NEP path; FP64 positions, velocities and accumulated per-atom outputs", "data": "synthetic",
            "config": {"workload": "PbTe %d atoms/GPU (replicate %d %d %d of the 250-atom cell per GPU), NEP NVE, dt 1 fs, 300 K"
                                   % ((n,) + reps),
                       "atoms_total": total, "parallelism": "spatial decomposition %dx%dx%d, ghost shell 2(rc+skin), "
                       "RCCL send/recv of ghost positions" % grid,
                       "local_atoms_max": int(n_loc.item()), "decompositions_in_timed_region": md.num_decompositions - dec0,
                       "mean_nn_radial": st.mean_nn_radial, "mean_nn_angular": st.mean_nn_angular},
            "roofline": roofline, "step_algorithmic_bytes_per_atom": b_step,
            "step_hbm_frac": b_step * (total * args.steps / elapsed) / (HBM_PEAK_GBS * 1e9 * world),
            "kernels": kern, "thermo_last": [float(v) for v in th],
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, nargs=3, default=[16, 16, 16], help="replicate na nb nc of the 250-atom cell")
    ap.add_argument("--workload", default="pbte", choices=["pbte", "carbon", "unep"],
                    help="pbte = BASELINE config 3 (the bench line); the others are extra single-GPU measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decomposed", action="store_true",
                    help="run the N > 1 code path (DomainMD) even on one GPU, to measure its host-side overhead")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:  # one instance of the all-cores CPU aggregate, no GPU involved
        cpu_worker(args.cpu_worker)
        return

    import torch
    import torch.distributed as dist
    import gpumd_amd
    import helpers as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    # NEPMI_DIST_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than
    # ranks (ranks share devices, messages staged through host memory); never a performance run.
    backend = os.environ.get("NEPMI_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # a wedged exchange should fail fast (watchdog), not hold the node for the default 10 min
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend)

    # ---- workload: every rank generates its own block of `reps` cells (weak scaling) ----
    reps = tuple(args.reps)
    label, nep_txt, h, typ, x, mass, vel = build_workload(args.workload, reps, 42 + rank)
    n = len(typ)
    model = gpumd_amd.Model(nep_txt)
    dt = 1.0 / H.TIME_UNIT
    if world > 1 or args.decomposed:
        run_decomposed(args, world, rank, dev, model, h, typ, x, mass, vel, reps)
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
    # initial force (Run::perform_a_run computes it before the loop), then warm-up steps
    eng.force_compute(h, t_type, t_x, t_pe, t_f, t_w)
    if args.warmup > 0:
        eng.run_nve(h, t_type, t_mass, dt, args.warmup, t_x, t_v, t_pe, t_f, t_w)
    # Inside the timed region only the dominant kernel (force assembly) carries HIP events -- two records per step; a
    # record around EVERY kernel stops them from running back to back and costs about 5 % of the step.  The full
    # per-kernel table comes from an instrumented pass of further steps after the clock has stopped.
    eng.set_timing(2)
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
        kern, roofline, b_step = kernel_report(st, model.info, n, st_all)
        out = {
            "metric": "atom-steps/sec, NEP PbTe NVE (nep4 2 Te Pb, examples/nep_train/nep.txt)",
            "value": value, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "dtype_note": "FP32 kernel arithmetic like the reference NEP path; FP64 positions, velocities and accumulated per-atom outputs", "data": "synthetic",
            "config": {"workload": label,
                       "atoms_total": total_atoms, "rebuilds_in_timed_region": int(st.num_rebuild - reb0),
                       "mean_nn_radial": st.mean_nn_radial, "mean_nn_angular": st.mean_nn_angular,
                       "lds_window_mode": int(st.radial_tiles), "parallelism": "1 GPU"},
            "roofline": roofline,
            "step_algorithmic_bytes_per_atom": b_step,
            "step_hbm_frac": b_step * (n * args.steps / elapsed) / (HBM_PEAK_GBS * 1e9),
            "kernels": kern,
            "kernels_note": "force_assemble: HIP events inside the timed region; the others: an instrumented pass of %d further steps" % extra,
            "thermo_last": [float(v) for v in th[-1]] if len(th) else None,
        }
        if args.workload == "pbte":
            # the path is FP32-VALU/gather bound, so the step is also priced against the
            # FP32 vector peak with the survey's FLOP count of the reference algorithm (8e4 per atom-step)
            out["fp32_valu"] = {"flop_per_atom_step_survey": 8.0e4, "equivalent_tflops": value * 8.0e4 / 1e12,
                                "peak_tflops": 157.3, "frac": value * 8.0e4 / 157.3e12,
                                "note": "the engine's own algebra needs ~3e4 FLOP per atom-step (DESIGN.md section 3)"}
        if world == 1 and not args.no_cpu_baseline and args.workload == "pbte":
            with _stdout_to_stderr():
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
