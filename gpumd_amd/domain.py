"""Spatial domain decomposition of the NEP NVE path over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  Replaces the reference's single-process slab scheme NEP_MULTIGPU
(src/force/nep_multigpu.cu:1416-1803: GPU 0 owns the state, scatters slab+halo positions and
gathers forces through blocking peer copies every step, integrator on GPU 0) with persistent
ownership:

  * a Cartesian process grid over the fractional coordinates of the global box; every rank owns the
    atoms inside its sub-box AND their integrator state;
  * ghost shell of width 2 (rc + skin): the reference's geometry (nep_multigpu.cuh:42-50) --
    positions only travel (forward communication), descriptors of the inner ring (rc + skin) are
    recomputed redundantly, forces are produced for owned atoms only, no reverse communication;
  * per step: ONE ghost-position exchange, staged over the decomposed directions (2 messages per
    direction; edges and corners are forwarded, so 6 messages instead of 26), as a grouped
    isend/irecv batch, plus one 1-int all-reduce for the re-decomposition vote;
  * migration + ghost-list rebuild only when some atom moved more than skin/2 since the last
    decomposition (the Verlet-skin criterion of src/force/neighbor.cu:741-800, made global).

The force call on the local (owned + ghost) system is nepmi_potential_compute_levels of the C ABI.
Everything here is device-agnostic torch code; the engine is injected.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

K_B = 8.617343e-5
SKIN = 1.0  # neighbor.cuh:212


def choose_grid(world):
    """Process grid for `world` ranks: as cubic as possible (all 7 peers of a 2x2x2 grid are
    direct xGMI links on an 8-GPU node)."""
    best = (world, 1, 1)
    for a in range(1, world + 1):
        if world % a:
            continue
        for b in range(1, world // a + 1):
            if (world // a) % b:
                continue
            c = world // a // b
            g = tuple(sorted((a, b, c), reverse=True))
            if max(g) - min(g) < max(best) - min(best):
                best = g
    return best


class DomainMD:
    def __init__(self, make_engine, rc, h9, pbc, grid, rank, world, device, group=None, stage_through_host=False,
                 overlap=None):
        """make_engine(capacity) -> gpumd_amd.NEP-like object.  h9: global Box::cpu_h[0..8].
        stage_through_host: move message payloads through host memory (only for running several
        ranks over gloo with device-resident state, e.g. two ranks sharing the one GPU of a test
        box); on a real node the payloads stay in HBM and travel over RCCL/xGMI."""
        self.make_engine = make_engine
        self.stage = bool(stage_through_host)
        # overlap the per-step ghost exchange with the interior radial pass (engine compute_levels_begin/_end);
        # NEPMI_OVERLAP=0 selects the plain exchange-then-compute order
        self.overlap = (os.environ.get("NEPMI_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self.side = None
        self.num_overlapped = 0
        self.plain_calls = 0
        self.rc = float(rc)
        self.dev = device
        self.rank, self.world, self.group = rank, world, group
        self.grid = tuple(int(g) for g in grid)
        assert self.grid[0] * self.grid[1] * self.grid[2] == world
        self.pbc = tuple(int(p) for p in pbc)
        self.h9 = np.asarray(h9, dtype=np.float64).reshape(9)
        H = self.h9.reshape(3, 3)  # columns a, b, c
        self.H = torch.tensor(H, dtype=torch.float64, device=device)
        self.G = torch.linalg.inv(self.H)
        vol = abs(np.linalg.det(H))
        a, b, c = H[:, 0], H[:, 1], H[:, 2]
        self.volume = vol
        self.thick = np.array([vol / np.linalg.norm(np.cross(b, c)), vol / np.linalg.norm(np.cross(c, a)),
                               vol / np.linalg.norm(np.cross(a, b))])
        self.coords = (rank % self.grid[0], (rank // self.grid[0]) % self.grid[1], rank // (self.grid[0] * self.grid[1]))
        self.lo = np.array([self.coords[d] / self.grid[d] for d in range(3)])
        self.hi = np.array([(self.coords[d] + 1) / self.grid[d] for d in range(3)])
        self.wfrac = (2.0 * self.rc + 2.0 * SKIN) / self.thick   # ghost shell
        self.ifrac = (self.rc + SKIN) / self.thick               # inner ring (descriptors)
        self.dims = [d for d in range(3) if self.grid[d] > 1]
        for d in self.dims:
            if self.wfrac[d] > 1.0 / self.grid[d]:
                raise ValueError("domain thinner than the ghost shell 2(rc+skin) in direction %d" % d)
        # local box handed to the engine: ghost-padded sub-box, open in the decomposed directions
        ext = np.ones(3)
        org = np.zeros(3)
        for d in self.dims:
            ext[d] = (self.hi[d] - self.lo[d]) + 2.0 * self.wfrac[d]
            org[d] = self.lo[d] - self.wfrac[d]
        self.h_loc = (H * ext[None, :]).reshape(9).copy()
        self.pbc_loc = tuple(0 if d in self.dims else self.pbc[d] for d in range(3))
        self.origin = torch.tensor(H @ org, dtype=torch.float64, device=device)
        self.engine = None
        self.n_total = None
        self.num_decompositions = 0

    # ---------------------------------------------------------------------------------------
    def rank_of(self, c):
        return c[0] + self.grid[0] * (c[1] + self.grid[1] * c[2])

    def _neighbor(self, d, step):
        c = list(self.coords)
        c[d] = (c[d] + step) % self.grid[d]
        return self.rank_of(c)

    def _p2p(self, sends, recvs):
        """sends/recvs: lists of (tensor, peer).  One grouped isend/irecv batch (ncclGroup on RCCL)."""
        if not sends and not recvs:
            return
        if self.stage:
            s_host = [(t.cpu(), p) for t, p in sends]
            r_host = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
        else:
            s_host, r_host = sends, recvs
        ops = [dist.P2POp(dist.isend, t, p, group=self.group) for t, p in s_host]
        ops += [dist.P2POp(dist.irecv, t, p, group=self.group) for t, p in r_host]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.stage:
            for (t, _), (th, _) in zip(recvs, r_host):
                t.copy_(th)

    def _all_reduce(self, t, op=None):
        op = op if op is not None else dist.ReduceOp.SUM
        if self.stage:
            th = t.cpu()
            dist.all_reduce(th, op=op, group=self.group)
            t.copy_(th)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    def _all_gather(self, t):
        th = t.cpu() if self.stage else t
        out = [torch.zeros_like(th) for _ in range(self.world)]
        dist.all_gather(out, th, group=self.group)
        return [o.to(self.dev) for o in out]

    # ---------------------------------------------------------------------------------------
    def setup(self, x, v, typ, mass, ids=None):
        """x, v: (3, n) float64 tensors of the atoms this rank starts with (any position: they are
        migrated to their owners); typ int32 (n), mass float64 (n)."""
        n = x.shape[1]
        if ids is None:
            me = torch.tensor([n], dtype=torch.int64, device=self.dev)
            if self.world > 1:
                counts = self._all_gather(me)
                base = int(sum(int(c.item()) for c in counts[: self.rank]))
            else:
                base = 0
            ids = torch.arange(base, base + n, dtype=torch.float64, device=self.dev)
        self.x, self.v = x.contiguous().clone(), v.contiguous().clone()
        self.typ, self.mass, self.ids = typ.clone(), mass.clone(), ids.clone().to(torch.float64)
        tot = torch.tensor([n], dtype=torch.int64, device=self.dev)
        if self.world > 1:
            self._all_reduce(tot)
        self.n_total = int(tot.item())
        self.decompose()

    # ---------------------------------------------------------------------------------------
    def _migrate(self):
        """Wrap owned atoms into the global cell and send each to the rank whose sub-box holds it."""
        s = self.G @ self.x
        for d in range(3):
            if self.pbc[d]:
                s[d] -= torch.floor(s[d])
        self.x = self.H @ s
        if self.world == 1:
            return
        dest = torch.zeros(self.x.shape[1], dtype=torch.int64, device=self.dev)
        mult = 1
        for d in range(3):
            cd = torch.clamp(torch.floor(s[d] * self.grid[d]).to(torch.int64), 0, self.grid[d] - 1)
            dest += cd * mult
            mult *= self.grid[d]
        payload = torch.cat([self.x, self.v, self.mass[None], self.typ.to(torch.float64)[None], self.ids[None]], 0)  # (9, n)
        counts = torch.bincount(dest, minlength=self.world)
        allc = torch.stack(self._all_gather(counts)).cpu().numpy()  # allc[src][dst]
        keep = payload[:, dest == self.rank]
        sends, recvs, recv_bufs = [], [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            if allc[self.rank][r] > 0:
                sends.append((payload[:, dest == r].contiguous(), r))
            if allc[r][self.rank] > 0:
                rb = torch.empty((9, int(allc[r][self.rank])), dtype=torch.float64, device=self.dev)
                recv_bufs.append(rb)
                recvs.append((rb, r))
        self._p2p(sends, recvs)
        new = torch.cat([keep] + recv_bufs, 1)
        # deterministic local order: ascending global id
        order = torch.argsort(new[8])
        new = new[:, order]
        self.x, self.v = new[0:3].contiguous(), new[3:6].contiguous()
        self.mass = new[6].contiguous()
        self.typ = new[7].to(torch.int32).contiguous()
        self.ids = new[8].contiguous()

    def decompose(self):
        """Migration + ghost construction (staged over the decomposed directions)."""
        self._migrate()
        n_own = self.x.shape[1]
        self.n_own = n_own
        x_loc = self.x
        t_loc = self.typ
        self.stages = []
        for d in self.dims:
            s_d = (self.G[d] @ x_loc)
            stage = {"d": d, "sends": [], "recvs": []}
            send_data = []
            for direction in (-1, +1):  # to the lower, then to the upper neighbour
                edge = (self.coords[d] == 0) if direction < 0 else (self.coords[d] == self.grid[d] - 1)
                if edge and not self.pbc[d]:
                    idx = torch.zeros(0, dtype=torch.int64, device=self.dev)
                elif direction < 0:
                    idx = torch.nonzero(s_d < self.lo[d] + self.wfrac[d]).flatten()
                else:
                    idx = torch.nonzero(s_d >= self.hi[d] - self.wfrac[d]).flatten()
                shift = torch.zeros(3, dtype=torch.float64, device=self.dev)
                if edge and self.pbc[d]:
                    shift = self.H[:, d] * (1.0 if direction < 0 else -1.0)
                peer = self._neighbor(d, direction)
                stage["sends"].append({"idx": idx, "shift": shift, "peer": peer})
                send_data.append(torch.cat([x_loc[:, idx] + shift[:, None], t_loc[idx].to(torch.float64)[None]], 0).contiguous())
            # counts, then payload.  Receive order: from the upper neighbour (its "to lower"), then
            # from the lower one -- matches the peers' send order when both neighbours are one rank.
            cnt_send = [torch.tensor([sd.shape[1]], dtype=torch.int64, device=self.dev) for sd in send_data]
            cnt_recv = [torch.zeros(1, dtype=torch.int64, device=self.dev) for _ in range(2)]
            peers_recv = [self._neighbor(d, +1), self._neighbor(d, -1)]
            self._p2p([(cnt_send[k], stage["sends"][k]["peer"]) for k in range(2)],
                      [(cnt_recv[k], peers_recv[k]) for k in range(2)])
            rbufs = [torch.empty((4, int(cnt_recv[k].item())), dtype=torch.float64, device=self.dev) for k in range(2)]
            self._p2p([(send_data[k], stage["sends"][k]["peer"]) for k in range(2) if send_data[k].shape[1] > 0],
                      [(rbufs[k], peers_recv[k]) for k in range(2) if rbufs[k].shape[1] > 0])
            off = x_loc.shape[1]
            for k in range(2):
                stage["recvs"].append({"peer": peers_recv[k], "off": off, "cnt": rbufs[k].shape[1]})
                off += rbufs[k].shape[1]
            x_loc = torch.cat([x_loc, rbufs[0][0:3], rbufs[1][0:3]], 1)
            t_loc = torch.cat([t_loc, rbufs[0][3].to(torch.int32), rbufs[1][3].to(torch.int32)])
            self.stages.append(stage)
        self.n_loc = x_loc.shape[1]
        self.x_loc = x_loc.contiguous()
        self.t_loc = t_loc.contiguous()
        # levels: 2 owned, 1 inside the inner ring (rc + skin box distance), 0 outer ghost
        lvl = torch.ones(self.n_loc, dtype=torch.int8, device=self.dev)
        s = self.G @ self.x_loc
        for d in self.dims:
            out = torch.clamp(torch.maximum(self.lo[d] - s[d], s[d] - self.hi[d]), min=0.0)
            lvl[out > self.ifrac[d]] = 0
        lvl[: n_own] = 2
        self.level = lvl.contiguous()
        self.x_ref = self.x.clone()
        # engine + per-decomposition work arrays
        if self.engine is None or self.n_loc > self.capacity:
            self.capacity = int(self.n_loc * 1.15) + 1024
            self.engine = self.make_engine(self.capacity)
            # the 0.5 A displacement vote below is the skin policy; the engine need not read its own flag back
            self.engine.set_external_skin(True)
            self.plain_calls = 0
        self.engine.invalidate()
        n = self.n_loc
        self.pe_loc = torch.zeros(n, dtype=torch.float64, device=self.dev)
        self.f_loc = torch.zeros((3, n), dtype=torch.float64, device=self.dev)
        self.w_loc = torch.zeros((9, n), dtype=torch.float64, device=self.dev)
        self.f_own = torch.zeros((3, n_own), dtype=torch.float64, device=self.dev)
        self.x_eng = torch.zeros((3, n), dtype=torch.float64, device=self.dev)
        self.num_decompositions += 1

    # ---------------------------------------------------------------------------------------
    def halo_update(self):
        """Forward communication of ghost positions (every step)."""
        self.x_loc[:, : self.n_own] = self.x
        for stage in self.stages:
            sbufs = [(self.x_loc[:, s["idx"]] + s["shift"][:, None]).contiguous() for s in stage["sends"]]
            rbufs = [torch.empty((3, r["cnt"]), dtype=torch.float64, device=self.dev) for r in stage["recvs"]]
            self._p2p([(sbufs[k], stage["sends"][k]["peer"]) for k in range(2) if sbufs[k].shape[1] > 0],
                      [(rbufs[k], stage["recvs"][k]["peer"]) for k in range(2) if rbufs[k].shape[1] > 0])
            for k in range(2):
                r = stage["recvs"][k]
                if r["cnt"]:
                    self.x_loc[:, r["off"]: r["off"] + r["cnt"]] = rbufs[k]

    def compute_forces(self):
        """Force::compute on the local system -> self.f_own (3, n_own), self.pe_loc, self.w_loc."""
        x_eng = (self.x_loc - self.origin[:, None]).contiguous()
        self.pe_loc.zero_()
        self.f_loc.zero_()
        self.w_loc.zero_()
        self.engine.compute_levels(self.h_loc, self.pbc_loc, self.n_loc, self.t_loc, x_eng, self.level,
                                   self.pe_loc, self.f_loc, self.w_loc)
        self.plain_calls += 1
        self.f_own.copy_(self.f_loc[:, : self.n_own])

    def halo_and_forces_overlapped(self):
        """halo_update + compute_forces with the exchange hidden behind the interior bricks' radial pass:
        the exchange runs on a side stream (its RCCL send/recv are ordered against that stream only),
        the engine's first half on the main stream touches owned atoms only, the second half waits
        for the exchange.  On the CPU tier there are no streams: the first half simply runs BEFORE the
        exchange, which proves it does not depend on this step's ghost positions."""
        n_own, e = self.n_own, self.engine
        self.x_eng[:, :n_own] = self.x - self.origin[:, None]
        self.pe_loc.zero_()
        self.f_loc.zero_()
        self.w_loc.zero_()
        args = (self.h_loc, self.pbc_loc, self.n_loc, self.t_loc, self.x_eng, self.level, self.pe_loc, self.f_loc,
                self.w_loc)
        if self.dev.type == "cuda":
            if self.side is None:
                self.side = torch.cuda.Stream(device=self.dev)
            main = torch.cuda.current_stream(self.dev)
            ready = torch.cuda.Event()
            ready.record(main)                      # owned positions (vv1) are final
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                self.halo_update()
                self.x_eng[:, n_own:] = self.x_loc[:, n_own:] - self.origin[:, None]
                done = torch.cuda.Event()
                done.record(self.side)
            started = e.compute_levels_begin(*args)
            main.wait_event(done)
        else:
            started = e.compute_levels_begin(*args)
            self.halo_update()
            self.x_eng[:, n_own:] = self.x_loc[:, n_own:] - self.origin[:, None]
        e.compute_levels_end(*args)
        self.num_overlapped += 1 if started else 0
        self.f_own.copy_(self.f_loc[:, :n_own])

    def needs_decomposition(self):
        d = self.x - self.x_ref
        flag = ((d * d).sum(0).max() > 0.25 * SKIN * SKIN).to(torch.int32).reshape(1)
        if self.world > 1:
            self._all_reduce(flag, dist.ReduceOp.MAX)
        return bool(flag.item())

    def initial_forces(self):
        self.halo_update()
        self.compute_forces()

    def step(self, dt):
        """One NVE step of Run::perform_a_run (run.cu:250-318) on the decomposed system."""
        e = self.engine
        e.lib.nepmi_vv_step1(e.handle, self.n_own, float(dt), e._ptr(self.mass), e._ptr(self.f_own),
                             e._ptr(self.x), e._ptr(self.v))
        if self.needs_decomposition():
            self.decompose()
            self.compute_forces()
        elif self.overlap and self.plain_calls >= 4:
            self.halo_and_forces_overlapped()
        else:
            # the engine's first four one-call evaluations also time its kernel variants (tile mode -1)
            self.halo_update()
            self.compute_forces()
        e = self.engine
        e.lib.nepmi_vv_step2(e.handle, self.n_own, float(dt), e._ptr(self.mass), e._ptr(self.f_own), e._ptr(self.v))

    def run(self, nsteps, dt):
        for _ in range(nsteps):
            self.step(dt)

    def thermo(self):
        """T, U, and the six stress components (find_thermo, ensemble.cu:434-673), global."""
        n = self.n_own
        m, v = self.mass, self.v
        w = self.w_loc[:, :n]
        parts = torch.stack([
            (m * (v * v).sum(0)).sum(), self.pe_loc[:n].sum(),
            w[0].sum() + (m * v[0] * v[0]).sum(), w[1].sum() + (m * v[1] * v[1]).sum(),
            w[2].sum() + (m * v[2] * v[2]).sum(), w[3].sum() + (m * v[0] * v[1]).sum(),
            w[4].sum() + (m * v[0] * v[2]).sum(), w[5].sum() + (m * v[1] * v[2]).sum()])
        if self.world > 1:
            self._all_reduce(parts)
        p = parts.cpu().numpy()
        out = np.empty(8)
        out[0] = p[0] / (3.0 * self.n_total * K_B)
        out[1] = p[1]
        out[2:] = p[2:] / self.volume
        return out

    def gather_owned(self):
        """(ids, x, v, f) of the owned atoms as numpy (for tests / dumps)."""
        return (self.ids.cpu().numpy().astype(np.int64), self.x.cpu().numpy().copy(), self.v.cpu().numpy().copy(),
                self.f_own.cpu().numpy().copy())
