"""The code path an 8-GPU node runs, on a 1-GPU box: the domain-decomposed driver with a DEVICE transport
(device_buffers = 1: ghost positions never leave the GPU, the skin vote is all-reduced on the device, steps are enqueued
speculatively behind the device flag) and MORE THAN ONE rank.  nepmi_transport_rccl cannot do that here (RCCL refuses two
ranks on one device) and the TCP transport of tests/test_dist.py is a host transport, so the ranks run as threads of one
process over tests/inproc/inproc_transport.hip (events + device-to-device copies on the ranks' own streams, the same
stream-ordered contract as the RCCL transport).  Compared with the single-domain run like tests/test_dist.py."""
import ctypes as C
import os
import tempfile
import threading

import numpy as np
import pytest

import helpers as H

LIB = {"gpu": os.path.join(H.ROOT, "tests", "inproc", "libinproc_transport.so"),
       "cpu": os.path.join(H.ROOT, "tests", "inproc", "libinproc_transport_host.so")}
HANG_SECONDS = 120  # a rank that has not finished by then is out of step with the others


def _run_threads(world, spec, expect_errors=False):
    import dist_worker
    from gpumd_amd import _capi
    from gpumd_amd.dist import Transport
    on_gpu = spec["device"] == "gpu"
    if on_gpu:
        import torch
    lib = C.CDLL(LIB[spec["device"]])
    lib.inproc_group_create.restype = C.c_void_p
    lib.inproc_group_create.argtypes = [C.c_int]
    lib.inproc_group_destroy.argtypes = [C.c_void_p]
    lib.inproc_transport.argtypes = [C.c_void_p, C.c_int, C.POINTER(_capi.NepmiTransport)]
    group = lib.inproc_group_create(world)
    assert group
    out = tempfile.mkdtemp(prefix="nepmi_inproc_")
    errors = []

    def body(rank):
        try:
            def make(drv):
                t = _capi.NepmiTransport()
                assert lib.inproc_transport(group, rank, C.byref(t)) == 0
                return Transport(drv.lib, t)
            stream = None
            if on_gpu:
                stream = torch.cuda.Stream()
                torch.cuda.synchronize()
            dist_worker.run_rank(out, spec, rank, world, make, stream=stream)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    import time
    t_end = time.time() + HANG_SECONDS
    for t in threads:
        t.join(timeout=max(1.0, t_end - time.time()))
    if any(t.is_alive() for t in threads):
        # daemon threads: the interpreter can still exit; do not wait for a stuck rank (GPU box time is budgeted)
        os.write(2, b"test_dist_inproc: a rank hangs (exchange / all-reduce sequence out of step)\n")
        os._exit(3)
    if expect_errors:
        return errors
    assert not errors, errors
    lib.inproc_group_destroy(group)
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


CASES = [
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0),      # re-decompositions under speculation
    (4, "PbTe-reps", (8, 2, 2), (4, 1, 1), "nve", 20, 3000.0),      # the slab grid of bench.py's weak scaling
    (4, "PbTe-reps", (4, 4, 2), (2, 2, 1), "nvt_ber", 16, 2000.0),
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nvt_nhc", 16, 2000.0),
    (2, "C-2022", (12, 6, 6), (2, 1, 1), "nvt_ber", 10, 3000.0),
]


def _check(device, world, model, reps, grid, ensemble, nsteps, temp, overlap=None, ghosts=None):
    if not os.path.exists(LIB[device]):
        pytest.skip("tests/inproc transports not built")
    import test_dist as T
    spec = T._spec(device, model, reps, grid, ensemble, nsteps, temp, ghosts=ghosts)
    if overlap is not None:
        spec["overlap"] = overlap
    n = T._natoms(model, reps)
    multi = _run_threads(world, spec)
    single = T._run_ranks(1, dict(spec, grid=[1, 1, 1]))
    xs, vs, fs, f0s = T._merge(single, n)
    xm, vm, fm, f0m = T._merge(multi, n)
    assert np.abs(f0m - f0s).max() < 3e-5 + 1e-5 * np.abs(f0s).max()
    assert np.abs(vm - vs).max() < 2e-6
    assert np.abs(fm - fs).max() < 30 * 3e-5 + 1e-4 * np.abs(fs).max()
    for r in multi:
        np.testing.assert_allclose(r["th1"], multi[0]["th1"], rtol=0, atol=0)
        np.testing.assert_allclose(r["th1"][:2], single[0]["th1"][:2], rtol=1e-6)
    if model == "PbTe-reps" and ensemble == "nve" and nsteps >= 20:
        assert max(int(r["ndec"]) for r in multi) >= 2
    if overlap:
        assert all(int(r["nover"]) > 0 for r in multi)
        # (these systems are too small for the one-lane window kernels, hence for the scatter form and its split force assembly:
        # test_reverse_exchange_overlapped_with_the_interior_bricks_on_gpu below)
    if ghosts is not None:
        assert all(int(r["reverse"]) == ghosts for r in multi)


# the interior / boundary split of the radial pass (nepmi_dist_set_overlap: boundary bricks on the communication stream behind
# the ghost unpack) and reverse-mode ghosts (nepmi_dist_set_ghost_mode) over the device transport
OPTION_CASES = [
    (2, "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 20, 3000.0, True, None),
    (4, "PbTe-reps", (4, 4, 2), (2, 2, 1), "nvt_nhc", 16, 2000.0, True, 1),
    (4, "PbTe-reps", (8, 2, 2), (4, 1, 1), "nve", 20, 3000.0, None, 1),
    (8, "PbTe-reps", (4, 4, 4), (2, 2, 2), "nve", 12, 3000.0, True, 1),
]


@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp,overlap,ghosts", OPTION_CASES[:3])
def test_device_transport_options_on_emulator(world, model, reps, grid, ensemble, nsteps, temp, overlap, ghosts):
    _check("cpu", world, model, reps, grid, ensemble, nsteps, temp, overlap, ghosts)


@pytest.mark.gpu
@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp,overlap,ghosts", OPTION_CASES)
def test_device_transport_options_on_gpu(world, model, reps, grid, ensemble, nsteps, temp, overlap, ghosts):
    _check("gpu", world, model, reps, grid, ensemble, nsteps, temp, overlap, ghosts)


@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp", CASES[:4])
def test_device_transport_with_several_ranks_on_emulator(world, model, reps, grid, ensemble, nsteps, temp):
    _check("cpu", world, model, reps, grid, ensemble, nsteps, temp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,model,reps,grid,ensemble,nsteps,temp", CASES)
def test_device_transport_with_several_ranks_on_gpu(world, model, reps, grid, ensemble, nsteps, temp):
    _check("gpu", world, model, reps, grid, ensemble, nsteps, temp)


@pytest.mark.gpu
def test_reverse_exchange_overlapped_with_the_interior_bricks_on_gpu():
    """Reverse-mode ghosts + nepmi_dist_set_overlap(1) on a system large enough for the scatter form of the force assembly
    (192,000 atoms on two ranks): the bricks whose window holds a ghost and the ghosts' fold first, the ghosts' partial forces
    on the communication stream while the interior bricks run -- bit-identical to the plain order (the same integer window
    sums, the returned parts added in the same order), and the split path is really taken."""
    if not os.path.exists(LIB["gpu"]):
        pytest.skip("tests/inproc transports not built")
    import test_dist as T
    spec = dict(T._spec("gpu", "PbTe-reps", (12, 8, 8), (2, 1, 1), "nve", 12, 1500.0, ghosts=1), force_form=1)
    a = _run_threads(2, dict(spec, overlap=True))
    b = _run_threads(2, dict(spec, overlap=False))
    assert all(int(r["nrev"]) >= 6 for r in a) and all(int(r["nrev"]) == 0 for r in b)
    assert all(int(r["reverse"]) == 1 for r in a)
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["i1"], rb["i1"])
        assert np.array_equal(ra["x1"], rb["x1"]) and np.array_equal(ra["v1"], rb["v1"]) and np.array_equal(ra["f1"], rb["f1"])
        np.testing.assert_allclose(ra["th1"], rb["th1"], rtol=0, atol=0)


@pytest.mark.gpu
def test_scatter_guard_band_in_a_decomposed_run_is_voted():
    """The fixed-point sums of the scatter-form force assembly in a decomposed run (192,000 atoms, two ranks, device transport,
    speculative enqueue).  The guard band is lowered on ONE rank only, so that ordinary forces trip it there: (a) the per-call
    evaluation (nepmi_dist_compute) is repeated in the gather form on BOTH ranks -- initial forces equal the gather-form run's bit
    for bit; (b) with the band lowered after that evaluation the loop's first steps trip it: the flag travels with the skin vote,
    both ranks leave the form together, the flagged steps stand, and the trajectory agrees with the gather-form run to the
    rounding of the two forms; (c) a band so low that the hard limit (four times the band) is reached ends the run with an error
    on both ranks instead of letting a wrapped sum through."""
    if not os.path.exists(LIB["gpu"]):
        pytest.skip("tests/inproc transports not built")
    import test_dist as T
    base = T._spec("gpu", "PbTe-reps", (12, 8, 8), (2, 1, 1), "nve", 24, 600.0, ghosts=1)
    gather = _run_threads(2, dict(base, force_form=0))
    plain = _run_threads(2, dict(base, force_form=1))
    assert all(int(r["nhand"]) == 0 for r in gather + plain)
    for a, b in zip(plain, gather):  # the two forms: the same trajectory to their rounding
        assert np.array_equal(a["i1"], b["i1"]) and np.abs(a["x1"] - b["x1"]).max() < 2e-6
    # (a) band 0.25 eV/A on rank 1 from the start (forces of this structure: a few eV/A): nepmi_dist_compute trips it there
    low = _run_threads(2, dict(base, force_form=1, scatter_guard=0.25, guard_ranks=[1], guard_hard_factor=1000.0))
    assert all(int(r["nhand"]) == 1 for r in low), [int(r["nhand"]) for r in low]
    for a, b in zip(low, gather):
        assert np.array_equal(a["i0"], b["i0"]) and np.array_equal(a["f0"], b["f0"])
        assert np.abs(a["x1"] - b["x1"]).max() < 2e-6 and np.abs(a["f1"] - b["f1"]).max() < 1e-3  # ... and so is the run
    # (b) the same band set AFTER the per-call evaluation: the loop's first steps trip it on rank 1, the vote carries it
    voted = _run_threads(2, dict(base, force_form=1, scatter_guard=0.25, guard_ranks=[1], guard_hard_factor=1000.0,
                                 guard_after_compute=True))
    assert all(int(r["nhand"]) == 1 for r in voted), [int(r["nhand"]) for r in voted]
    for a, b in zip(voted, gather):
        assert np.array_equal(a["i1"], b["i1"])
        assert np.abs(a["x1"] - b["x1"]).max() < 2e-6 and np.abs(a["f1"] - b["f1"]).max() < 1e-3
    # (d) the band tripped on rank 1 DURING A STEP THE HOST LOOKS AT (the run's fourth force assembly: step 3, (3 + 1) % 4 == 0), hot
    # enough for list rebuilds inside the speculation window: the look is taken behind the vote and in front of the force phase,
    # so both ranks see the flag at the same look (a rank acting on its own, un-voted flag one look early would re-decompose or
    # throw alone and the other would wait for ever in its next collective: this case then ends in the harness's hang detector)
    hot = dict(T._spec("gpu", "PbTe-reps", (12, 8, 8), (2, 1, 1), "nve", 28, 3000.0, ghosts=1), thermo_every=7)  # (looks at steps 3, 7, 11, ...)
    hot_gather = _run_threads(2, dict(hot, force_form=0))
    hot_voted = _run_threads(2, dict(hot, force_form=1, scatter_guard=0.25, guard_ranks=[1], guard_hard_factor=1000.0,
                                     guard_after_compute=True, guard_delay=4))
    assert all(int(r["nhand"]) == 1 for r in hot_voted), [int(r["nhand"]) for r in hot_voted]
    assert max(int(r["ndec"]) for r in hot_voted) >= 2  # list rebuilds (re-decompositions) happened inside the run
    for a, b in zip(hot_voted, hot_gather):
        assert np.array_equal(a["i1"], b["i1"])
        assert np.abs(a["x1"] - b["x1"]).max() < 2e-5 and np.abs(a["f1"] - b["f1"]).max() < 1e-2
    # (c) the hard limit (default: four times the band): band 0.01 eV/A after compute() -> an error on every rank
    errors = _run_threads(2, dict(base, force_form=1, scatter_guard=0.01, guard_after_compute=True), expect_errors=True)
    assert sorted(r for r, _ in errors) == [0, 1], errors
    assert all("fixed-point" in msg for _, msg in errors), errors


def test_a_capacity_error_of_one_rank_ends_the_run_on_every_rank():
    """40 steps at 3000 K melt the PbTe block until one rank exceeds the angular list capacity of the model file (MN): the
    capacity bits travel with the skin vote, so both ranks report the error at the same point instead of one of them
    waiting for ever in its next collective."""
    import test_dist as T
    if not os.path.exists(LIB["cpu"]):
        pytest.skip("tests/inproc transports not built")
    spec = T._spec("cpu", "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 40, 3000.0)
    errors = _run_threads(2, spec, expect_errors=True)
    assert sorted(r for r, _ in errors) == [0, 1], errors
    assert all("capacity" in msg for _, msg in errors), errors


def test_an_atom_id_outside_the_system_is_refused_on_every_rank():
    """nepmi_dist_setup: the caller's ids label the gathered output and seed the Langevin generators; one id outside
    0 .. n_total - 1 on ONE rank is reported by both (the check is all-reduced), before anything is decomposed."""
    import test_dist as T
    if not os.path.exists(LIB["cpu"]):
        pytest.skip("tests/inproc transports not built")
    spec = T._spec("cpu", "PbTe-reps", (4, 2, 2), (2, 1, 1), "nve", 2, 300.0)
    spec["bad_ids"] = True
    errors = _run_threads(2, spec, expect_errors=True)
    assert sorted(r for r, _ in errors) == [0, 1], errors
    assert all("atom ids must be" in msg for _, msg in errors), errors
