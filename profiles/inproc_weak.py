#!/usr/bin/env python3
"""Decomposition overhead measured on ONE GPU: R ranks (threads over the in-process device transport of tests/inproc,
the RCCL transport's contract) each own a 1,024,000-atom PbTe block of an R x 1 x 1 slab arrangement and step
concurrently on the same MI355X; the same R x 1,024,000 atoms as ONE domain are the comparison.  Compute is shared, so
    overhead = t(R ranks) / t(one domain of R blocks)
is what the ghost atoms (positions received, descriptors of the inner ring recomputed), the exchange and the vote add
per owned atom -- everything of an R-GPU weak-scaling step except the xGMI transfer time itself.

    python profiles/inproc_weak.py [--ranks 2] [--steps 60] [--reps 16 16 16]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, nargs=3, default=[16, 16, 16])
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE block of --reps cells shared by the ranks (most cubic grid) against the same block as "
                         "one domain -- the work the decomposition adds when a fixed system is cut into R sub-boxes")
    ap.add_argument("--workload", default="pbte", choices=["pbte", "carbon", "unep"],
                    help="pbte: --reps of the 250-atom cell; carbon / unep: --reps conventional cells of diamond (C_2022) / the fcc 16-metal alloy (UNEP-v1)")
    ap.add_argument("--ghosts", type=int, default=-1,
                    help="nepmi_dist_set_ghost_mode: -1 the counted rule, 0 forward (shell 2 (rc + skin)), 1 reverse (shell rc + skin, "
                         "the ghosts' partial forces return to the owners)")
    ap.add_argument("--overlap", type=int, default=-1, help="nepmi_dist_set_overlap (interior bricks' radial pass before the ghosts arrive): 0 / 1, -1 = the library's default")
    ap.add_argument("--only", default="both", choices=["both", "ranks", "one"], help="profiling: run only one of the two legs")
    args = ap.parse_args()
    import torch
    import gpumd_amd
    from gpumd_amd import _capi, structures as S
    from gpumd_amd.dist import DistMD, Transport
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = gpumd_amd.load_library()
    tlib = C.CDLL(os.path.join(ROOT, "tests", "inproc", "libinproc_transport.so"))
    tlib.inproc_group_create.restype = C.c_void_p
    tlib.inproc_group_create.argtypes = [C.c_int]
    tlib.inproc_transport.argtypes = [C.c_void_p, C.c_int, C.POINTER(_capi.NepmiTransport)]
    model = gpumd_amd.Model(S.golden({"pbte": "PbTe", "carbon": "C", "unep": "UNEP"}[args.workload], "nep.txt"))
    dt = 1.0 / S.TIME_UNIT
    R = args.ranks
    reps = tuple(args.reps)

    def block(rank):
        if args.workload == "carbon":
            h, typ, x, mass, vel = S.diamond_block(reps, seed=42 + rank)
        elif args.workload == "unep":
            h, typ, x, mass, vel = S.fcc_alloy_block(reps, seed=42 + rank)
        else:
            h, typ, x, mass, vel = S.pbte_block(reps, seed=42 + rank)
        return np.asarray(h).reshape(3, 3), typ, x, mass, vel

    def run(world, blocks_per_rank):
        """world ranks; each owns blocks_per_rank consecutive blocks of the R-block slab system"""
        group = tlib.inproc_group_create(world)
        barrier = threading.Barrier(world)
        times, infos, errs = [0.0] * world, [None] * world, []

        def body(rank):
            try:
                Hb = block(0)[0]
                Hg = Hb * np.asarray([R, 1, 1], dtype=np.float64)[None, :]
                Xs, Vs, Ts, Ms = [], [], [], []
                for b in range(rank * blocks_per_rank, (rank + 1) * blocks_per_rank):
                    _, typ, x, mass, vel = block(b)
                    n = len(typ)
                    Xs.append(x.reshape(3, n) + (Hb @ np.asarray([b, 0, 0], dtype=np.float64))[:, None])
                    Vs.append(vel.reshape(3, n)); Ts.append(typ); Ms.append(mass)
                X, V = np.concatenate(Xs, axis=1), np.concatenate(Vs, axis=1)
                T, M = np.concatenate(Ts), np.concatenate(Ms)
                t = _capi.NepmiTransport()
                assert tlib.inproc_transport(group, rank, C.byref(t)) == 0
                tr = Transport(lib, t)
                stream = torch.cuda.Stream()
                md = DistMD(model, tr, Hg.reshape(9), (1, 1, 1), (world, 1, 1), stream=stream, ghost_mode=args.ghosts)
                md.setup(torch.from_numpy(np.ascontiguousarray(T)).to(dev), torch.from_numpy(np.ascontiguousarray(M)).to(dev),
                         torch.from_numpy(np.ascontiguousarray(X).reshape(-1)).to(dev),
                         torch.from_numpy(np.ascontiguousarray(V).reshape(-1)).to(dev))
                torch.cuda.synchronize()
                if args.overlap >= 0:
                    md.set_overlap(bool(args.overlap))
                md.compute()
                md.run("nve", dt, args.warmup)
                torch.cuda.synchronize()
                barrier.wait()
                t0 = time.perf_counter()
                md.run("nve", dt, args.steps)
                torch.cuda.synchronize()
                barrier.wait()
                times[rank] = time.perf_counter() - t0
                i = md.info()
                infos[rank] = (int(i.n_owned), int(i.n_local), int(i.num_decompositions), float(i.decompose_ms), int(i.reverse_ghosts))
                md.close()
                tr.close()
            except BaseException as e:  # noqa: BLE001
                errs.append((rank, repr(e)))
                try:
                    barrier.abort()
                except Exception:
                    pass

        ths = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in ths]
        t_end = time.time() + 240
        for t in ths:
            t.join(timeout=max(1.0, t_end - time.time()))
        if any(t.is_alive() for t in ths) or errs:
            sys.stderr.write("inproc_weak: %s\n" % (errs or "a rank hangs"))
            os._exit(3)
        return max(times), infos

    def run_strong(world):
        from gpumd_amd.dist import choose_grid
        group = tlib.inproc_group_create(world)
        barrier = threading.Barrier(world)
        times, infos, errs = [0.0] * world, [None] * world, []
        Hb, typ, x, mass, vel = block(0)
        n = len(typ)
        grid = choose_grid(world)

        def body(rank):
            try:
                mine = np.arange(n) % world == rank
                t = _capi.NepmiTransport()
                assert tlib.inproc_transport(group, rank, C.byref(t)) == 0
                tr = Transport(lib, t)
                stream = torch.cuda.Stream()
                md = DistMD(model, tr, Hb.reshape(9), (1, 1, 1), grid, stream=stream, ghost_mode=args.ghosts)
                md.setup(torch.from_numpy(np.ascontiguousarray(typ[mine])).to(dev),
                         torch.from_numpy(np.ascontiguousarray(mass[mine])).to(dev),
                         torch.from_numpy(np.ascontiguousarray(x.reshape(3, n)[:, mine]).reshape(-1)).to(dev),
                         torch.from_numpy(np.ascontiguousarray(vel.reshape(3, n)[:, mine]).reshape(-1)).to(dev))
                torch.cuda.synchronize()
                if args.overlap >= 0:
                    md.set_overlap(bool(args.overlap))
                md.compute()
                md.run("nve", dt, args.warmup)
                torch.cuda.synchronize()
                barrier.wait()
                t0 = time.perf_counter()
                md.run("nve", dt, args.steps)
                torch.cuda.synchronize()
                barrier.wait()
                times[rank] = time.perf_counter() - t0
                i = md.info()
                infos[rank] = (int(i.n_owned), int(i.n_local), int(i.num_decompositions), float(i.decompose_ms), int(i.reverse_ghosts))
                md.close()
                tr.close()
            except BaseException as e:  # noqa: BLE001
                errs.append((rank, repr(e)))
                try:
                    barrier.abort()
                except Exception:
                    pass

        ths = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
        [t.start() for t in ths]
        t_end = time.time() + 240
        for t in ths:
            t.join(timeout=max(1.0, t_end - time.time()))
        if any(t.is_alive() for t in ths) or errs:
            sys.stderr.write("inproc_weak: %s\n" % (errs or "a rank hangs"))
            os._exit(3)
        return max(times), infos

    if args.strong and args.only != "both":
        t, info = run_strong(R if args.only == "ranks" else 1)
        print(json.dumps({"mode": "strong", "ms_per_step": t / args.steps * 1e3, "info": info}))
        return
    if args.only == "ranks":
        t_multi, info_multi = run(R, 1)
        print(json.dumps({"ms_per_step": t_multi / args.steps * 1e3, "info": info_multi}))
        return
    if args.only == "one":
        t_single, info_single = run(1, R)
        print(json.dumps({"ms_per_step": t_single / args.steps * 1e3, "info": info_single}))
        return
    if args.strong:
        t_multi, info_multi = run_strong(R)
        t_single, info_single = run_strong(1)
        print(json.dumps({"mode": "strong", "ranks": R, "reverse_ghosts": info_multi[0][4], "atoms_total": sum(i[0] for i in info_multi),
                          "owned_per_rank": [i[0] for i in info_multi], "local_per_rank": [i[1] for i in info_multi],
                          "ms_per_step_ranks_sharing_one_gpu": t_multi / args.steps * 1e3,
                          "ms_per_step_one_domain": t_single / args.steps * 1e3, "work_inflation": t_multi / t_single,
                          "decompositions": [i[2] for i in info_multi]}))
        return
    t_multi, info_multi = run(R, 1)
    t_single, info_single = run(1, R)
    n_total = sum(i[0] for i in info_multi)
    out = {"ranks": R, "reverse_ghosts": info_multi[0][4], "atoms_per_rank": info_multi[0][0], "local_atoms_per_rank": [i[1] for i in info_multi],
           "steps": args.steps, "ms_per_step_ranks_sharing_one_gpu": t_multi / args.steps * 1e3,
           "ms_per_step_one_domain_same_atoms": t_single / args.steps * 1e3,
           "decomposition_overhead": t_multi / t_single,
           "predicted_weak_scaling_efficiency_excl_xgmi": t_single / t_multi,
           "atoms_total": n_total, "decompositions": [i[2] for i in info_multi],
           "decompose_ms_total_incl_setup": [round(i[3], 2) for i in info_multi],
           "one_domain_decompositions": info_single[0][2], "one_domain_decompose_ms_total": round(info_single[0][3], 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
