"""Worker of tests/test_dist.py: one rank of the C++ domain-decomposed driver (nepmi_dist_*, include/nepmi.h) over
the TCP transport.  "cpu": kernel logic from the test-only emulator library; "gpu": the product library, all ranks
sharing cuda:0 of a 1-GPU test box (RCCL refuses two ranks on one device, so payloads are staged through the host)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import parity_cases as P  # noqa: E402
from gpumd_amd.dist import DistMD, Transport  # noqa: E402


def main():
    out_dir, spec = sys.argv[1], json.loads(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    run_rank(out_dir, spec, rank, world,
             lambda drv: Transport.tcp(drv.lib, "127.0.0.1", int(os.environ["MASTER_PORT"]), rank, world))


def run_rank(out_dir, spec, rank, world, make_transport, stream=None):
    """One rank of a decomposed run; make_transport(driver) -> gpumd_amd.dist.Transport (TCP between processes, or the
    in-process device transport of tests/inproc between threads)."""
    on_gpu = spec["device"] == "gpu"
    drv = H.GpuDriver() if on_gpu else H.EmuDriver()
    nep_rel, build, _ = P.MODELS[spec["model"]] if spec["model"] in P.MODELS else (None, None, None)
    if spec["model"] == "PbTe-reps":
        nep = H.golden("PbTe", "nep.txt")
        h, typ, x = H.pbte_supercell(tuple(spec["reps"]), rattle=0.02, seed=31)
        masses = {0: H.MASS["Te"], 1: H.MASS["Pb"]}
    elif spec["model"] == "C-2022":
        nep = H.golden("C", "nep.txt")
        h, typ, x = H.diamond(tuple(spec["reps"]), 3.57, rattle=0.02, seed=32)
        masses = {0: H.MASS["C"]}
    elif spec["model"] == "UNEP-v1":
        nep = H.golden("UNEP", "nep.txt")
        h, typ, x = H.fcc_alloy(tuple(spec["reps"]), 3.9, 16, rattle=0.02, seed=33)
        masses = {t: 60.0 + 7.0 * t for t in range(16)}
    else:
        raise SystemExit("unknown model")
    model = drv.model(nep)
    n = len(typ)
    typ = typ.astype(np.int32)
    mass = np.array([masses[int(t)] for t in typ], dtype=np.float64)
    vel = H.maxwell_velocities(mass, spec["temp"], seed=5)
    mine = np.arange(n) % world == rank  # arbitrary initial distribution; setup() migrates
    ids = np.arange(n, dtype=np.int64)[mine]
    if spec.get("bad_ids") and rank == world - 1:  # one id outside 0 .. n_total - 1, on one rank only
        ids[0] = n
    tr = make_transport(drv)
    md = DistMD(model, tr, h, tuple(spec.get("pbc", (1, 1, 1))), spec["grid"], stream=stream, ghost_mode=spec.get("ghosts"))
    md.setup(drv.dev(typ[mine]), drv.dev(mass[mine]), drv.dev(np.ascontiguousarray(x.reshape(3, n)[:, mine]).reshape(-1)),
             drv.dev(np.ascontiguousarray(vel.reshape(3, n)[:, mine]).reshape(-1)), drv.dev(ids))
    if "overlap" in spec:
        md.set_overlap(spec["overlap"])
    if spec.get("force_form") is not None:  # the local engine's force-assembly form (systems below the run loops' size rule)
        e = md.lib.nepmi_dist_engine(md.handle)
        md._ck(md.lib.nepmi_engine_set_force_form(e, int(spec["force_form"])))
        if int(spec["force_form"]) == 1:  # the scatter form is the one-lane form: pin it (the rule takes two lanes up to 512 bricks)
            md._ck(md.lib.nepmi_engine_set_option(e, b"win_lanes", 1.0))
    def set_guard():  # narrow the guard band of the scatter form (on the ranks listed, default: all)
        if spec.get("scatter_guard") is not None and (spec.get("guard_ranks") is None or rank in spec["guard_ranks"]):
            e = md.lib.nepmi_dist_engine(md.handle)
            if spec.get("guard_delay"):  # ... from the n-th force assembly of the run on
                md._ck(md.lib.nepmi_engine_set_option(e, b"scatter_guard_delay", float(spec["guard_delay"])))
            md._ck(md.lib.nepmi_engine_set_option(e, b"scatter_guard", float(spec["scatter_guard"])))
            md._ck(md.lib.nepmi_engine_set_option(e, b"scatter_guard_hard", float(spec.get("guard_hard_factor", 0.0))))
    if not spec.get("guard_after_compute"):
        set_guard()
    if spec.get("seed") is not None:
        md.bdp_seed(spec["seed"])
        md.lan_seed(spec["seed"])
    md.compute()
    if spec.get("guard_after_compute"):
        set_guard()

    def snapshot():
        no = md.info().n_owned
        d_id = drv.dev(np.zeros(no, dtype=np.int64))
        d_x, d_v, d_f = drv.zeros(3 * no), drv.zeros(3 * no), drv.zeros(3 * no)
        md.gather_owned(d_id, d_x, d_v, d_f)
        return drv.host(d_id), drv.host(d_x).reshape(3, no), drv.host(d_v).reshape(3, no), drv.host(d_f).reshape(3, no)

    i0, x0, v0, f0 = snapshot()
    th0 = md.thermo()
    dt = spec["dt_fs"] / H.TIME_UNIT
    th = md.run(spec["ensemble"], dt, spec["nsteps"], spec.get("t1", 0.0), spec.get("t2", 0.0), spec.get("tcoup", 1.0),
                thermo_every=spec.get("thermo_every", 0))
    i1, x1, v1, f1 = snapshot()
    th1 = md.thermo()
    # per-atom virials (reverse-mode ghosts: completed by one more reverse exchange), and the global sums afterwards
    no = md.info().n_owned
    d_w = drv.zeros(9 * no)
    md.gather_owned(None, None, None, None, None, d_w)
    w1 = drv.host(d_w).reshape(9, no)
    th2 = md.thermo()
    info = md.info()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), i0=i0, f0=f0, th0=th0, i1=i1, x1=x1, v1=v1, f1=f1, th1=th1, th=th, w1=w1, th2=th2,
             n_loc=info.n_local, n_own=info.n_owned, reverse=info.reverse_ghosts, ndec=info.num_decompositions, nover=info.num_overlapped,
             nrev=md.num_overlapped_reverse(), nhand=info.num_range_handovers)
    md.close()
    tr.close()


if __name__ == "__main__":
    main()
