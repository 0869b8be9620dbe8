"""bench.py's launch contract: `python bench.py --gpus N` with no launcher around it must start its own ranks
(the reference starts its multi-GPU path from one process: src/force/force.cu:122-160, nep_multigpu.cu:1416-1803) and
print exactly one JSON line; a launch that cannot run says where it failed in an {"error": ..., "stage": ...} line."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

BENCH = os.path.join(H.ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout, cwd=H.ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    return out, lines


def test_self_launch_without_a_gpu_reports_an_error_line():
    """CPU tier: no device -> one JSON line with "error" and the stage, exit code != 0 (never a silent death)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the error path of the device probe is not reachable")
    out, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout + out.stderr
    line = json.loads(lines[0])
    assert "error" in line and line["stage"] == "self-launch" and line["n_gpus"] == 2


@pytest.mark.gpu
def test_plain_gpus_2_launches_its_own_ranks_and_prints_one_line():
    """GPU tier (one GPU on the box): `bench.py --gpus 2` with no launcher -- the two ranks share the device over the TCP
    transport (the protocol of the 2-GPU run: decomposition, ghost exchange, skin vote, re-decomposition), weak-scaling
    line + the strong-scaling leg of the same system under extra_measurements.strong."""
    out, lines = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--reps", "6", "4", "4", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert "error" not in line, line
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6 and line["warmup"] == 2
    assert line["value"] > 0 and line["unit"] == "atom-steps/s"
    cfg = line["config"]
    assert cfg["atoms_total"] == 2 * 6 * 4 * 4 * 250
    assert cfg["ghost_mode"] in ("forward", "reverse") and cfg["local_atoms_max"] > 6 * 4 * 4 * 250
    assert len(cfg["per_rank_ms_per_step"]["all"]) == 2 and "2x1x1" in cfg["parallelism"]
    # the form probe before the clock: four combinations timed, the headline ran the one that won
    probe = cfg["form_probe_before_the_clock"]
    assert len(probe["results"]) == 4 and all("ms_per_step" in v for v in probe["results"].values()), probe
    best = min(probe["results"].values(), key=lambda v: v["ms_per_step"])
    assert cfg["overlap"] == best["overlap"] and cfg["ghost_mode"] == ("reverse" if best["ghosts"] else "forward")
    strong = line["extra_measurements"]["strong"]
    assert "error" not in strong, strong
    assert strong["scaling"] == "strong" and strong["config"]["atoms_total"] == 6 * 4 * 4 * 250 and strong["value"] > 0
    # the decomposed forces against a one-domain evaluation of the same positions on rank 0 (a wrong ghost would show here)
    assert strong["checksum"]["atoms"] == 6 * 4 * 4 * 250 and strong["checksum"]["max_abs_dF_eV_per_A"] < 3e-5, strong["checksum"]
    # exchange/compute overlap {0, 1} x ghosts {forward, reverse} on both legs, after the clock
    matrix = line["extra_measurements"]["overlap_x_ghosts"]
    assert len(matrix) == 8 and all("error" not in v and v["value"] > 0 for v in matrix.values()), matrix
    assert {v["ghost_mode"] for v in matrix.values()} == {"forward", "reverse"} and {v["overlap"] for v in matrix.values()} == {0, 1}
    import torch
    if torch.cuda.device_count() < 2:
        assert "functional_only" in line


@pytest.mark.gpu
def test_a_failing_rank_names_its_stage():
    """a transport that cannot come up (an unroutable rendezvous for the TCP mesh) -> {"error", "stage"} and rc != 0"""
    out, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reps", "4", "4", "4", "--no-cpu-baseline", "--workload",
                       "pbte", "--ghosts", "0"], env={"NEPMI_DIST_BACKEND": "tcp", "NEPMI_BENCH_FAIL_STAGE": "setup"})
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    line = json.loads(lines[0])
    assert "error" in line and "stage" in line and "setup" in line["stage"], line
