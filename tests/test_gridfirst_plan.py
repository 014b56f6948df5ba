"""Host-side plan of the grid-first elimination order (camera_calibration_amd/csrc/gridfirst_plan.h), checked on the CPU.

The plan is static structure: the elimination order of the grid unknowns (strips + separators), the layout of the full normal
matrix F = [grid | rig | points | poses | right-hand side], the pivot chains, and the task lists of the block-sparse dataflow
launch (which 64 x 64 tiles exist, over which earlier block rows each accumulates, in what order they are handed out).  These
tests replay the task lists with numpy -- the same block arithmetic the kernel k_ldlt_sparse does, executed by a small pool of
simulated workgroups that take tickets in list order and block until their inputs exist -- and compare the result with a dense
solve of the same system.  A contribution the plan forgot, a tile it dropped or an order that can deadlock shows up here, without
a GPU.  (Reference counterpart of the solve: LV/lm_optimizer.h:1247-1369; SURVEY.md fact 3: any exact order gives the same x.)
"""
from __future__ import annotations

import numpy as np
import pytest

from camera_calibration_amd import engine
from camera_calibration_amd.problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera

B = 64


def _cam(model, gw, gh):
    return Camera(model, 640, 480, 0, 0, 639, 479, gw, gh)


def _build_system(cams, N, P, plan, rng, lam=1e-2):
    """Random J^T J + lambda I with the structure of the BA problem, in the order of F; right-hand side in column n_pad - 1."""
    C = len(cams)
    n_pad, Gf, n_rp = plan["n_pad"], plan["Gf"], plan["n_rp"]
    rig = 6 * C if C > 1 else 0
    cam_first = np.cumsum([0] + [c.params_per_grid_point * c.grid_points for c in cams])
    nF = Gf + plan["n_border"]
    H = np.zeros((nF, nF))
    n_obs = 40 * N * C
    for _ in range(n_obs):
        c = int(rng.integers(C)); i = int(rng.integers(N)); p = int(rng.integers(P))
        cam = cams[c]; ppg = cam.params_per_grid_point
        cx = int(rng.integers(cam.grid_w - 3)); cy = int(rng.integers(cam.grid_h - 3))
        cols = []
        for yy in range(4):
            for xx in range(4):
                seq = (cx + xx) + (cy + yy) * cam.grid_w
                for d in range(ppg):
                    cols.append(int(plan["f_of_grid"][cam_first[c] + ppg * plan["gperm"][c][seq] + d]))
        cols += [Gf + n_rp + 6 * i + k for k in range(6)]
        cols += [Gf + rig + 3 * p + k for k in range(3)]
        if C > 1:
            cols += [Gf + 6 * c + k for k in range(6)]
        cols = np.array(cols)
        J = rng.normal(size=(2, cols.size))
        H[np.ix_(cols, cols)] += J.T @ J
    H += lam * np.mean(np.diag(H)[np.diag(H) > 0]) * np.eye(nF)
    F = np.zeros((n_pad, n_pad))
    F[:nF, :nF] = H
    # identity rows: padding inside the grid part (never touched by an observation) and behind the border
    real = np.zeros(n_pad, bool)
    real[plan["f_of_grid"]] = True
    real[Gf:nF] = True
    for j in np.nonzero(~real)[0]:
        F[j, :] = 0.0; F[:, j] = 0.0; F[j, j] = 1.0
    b = np.zeros(n_pad)
    b[real] = rng.normal(size=int(real.sum()))
    b[n_pad - 1] = 1.0                               # the last diagonal entry (the column doubles as the right-hand side)
    F[:, n_pad - 1] = b
    F[n_pad - 1, :] = 0.0
    F[n_pad - 1, n_pad - 1] = 1.0
    return np.triu(F), real


def _ldlt(T):
    """unpivoted LDL^T of a symmetric positive definite block"""
    Cc = np.linalg.cholesky(T)
    s = np.diag(Cc).copy()
    return Cc / s, s * s


class Replay:
    """Block arithmetic of the dataflow launch driven by the plan's task lists."""

    def __init__(self, plan, F):
        self.pl = plan
        self.Z = F.copy()                               # upper triangle, row-major: tile (r, c) = Z[64 r : 64 r + 64, 64 c : 64 c + 64]
        nbg, ntc = plan["nbg"], plan["ntc"]
        self.X = np.zeros((plan["Gf"], plan["n_pad"]))  # X = D L of the border columns (the B operand of the border update)
        self.tile = np.zeros((nbg, ntc), bool)
        self.diag = np.zeros(nbg, bool); self.upre = np.zeros(nbg, bool); self.part = np.zeros(nbg, bool)
        self.d = np.zeros((plan["n_fact"] // B, B)); self.invL = np.zeros((plan["n_fact"] // B, B, B))
        self.chain_pos = [int(c[0]) for c in plan["chains"]]

    def t(self, r, c):
        return self.Z[B * r:B * r + B, B * c:B * c + B]

    def task_ready(self, task):
        kind = task[0] & 255; n_iv = task[0] >> 8; r, c, iv0 = int(task[1]), int(task[2]), int(task[3])
        ca = c if kind in (1, 3) else r
        for k0, k1 in self.pl["ivals"][iv0:iv0 + n_iv]:
            for k in range(k0, k1):
                if not (self.tile[k, ca] and self.tile[k, c]):
                    return False
        if kind == 4:
            for k0, k1 in self.pl["ivals"][iv0:iv0 + n_iv]:
                for k in range(k0, k1):
                    if not self.tile[k, c + 1]:
                        return False
        if kind in (2, 4) and not self.diag[r]:
            return False
        return True

    def run_task(self, task):
        kind = task[0] & 255; n_iv = task[0] >> 8; r, c, iv0 = int(task[1]), int(task[2]), int(task[3])
        if kind == 4:                                    # REG2: the two tiles of one 128-column border tile, same K intervals
            for cc in (c, c + 1):
                self.run_task(np.array([2 | (n_iv << 8), r, cc, iv0]))
            return
        ca = c if kind in (1, 3) else r
        row = c if kind in (1, 3) else r
        acc = np.zeros((B, B))
        for k0, k1 in self.pl["ivals"][iv0:iv0 + n_iv]:
            for k in range(k0, k1):
                acc += (self.t(k, ca) * self.d[k][:, None]).T @ self.t(k, c)
        U = self.t(row, c) - acc
        if kind == 0:
            self.t(r, c)[:] = U; self.upre[r] = True
        elif kind in (1, 3):
            self.t(c, c)[:] = np.triu(U); self.part[c] = True
        else:
            Xv = self.invL[r] @ U
            self.t(r, c)[:] = Xv / self.d[r][:, None]
            if c >= self.pl["nbg"]:
                self.X[B * r:B * r + B, B * c:B * c + B] = Xv
            self.tile[r, c] = True

    def chain_step(self, i):
        """advances chain i by one block if its inputs exist"""
        r0, r1, dep, _ = (int(v) for v in self.pl["chains"][i])
        r = self.chain_pos[i]
        if r >= r1:
            return False
        if r == r0:
            if dep and not self.part[r]:
                return False
            T = self.t(r, r).copy()
        else:
            if not (self.upre[r - 1] and self.part[r]):
                return False
            Xv = self.invL[r - 1] @ self.t(r - 1, r)
            Lt = Xv / self.d[r - 1][:, None]
            self.t(r - 1, r)[:] = Lt
            self.tile[r - 1, r] = True
            T = self.t(r, r) - Lt.T @ Xv
        T = np.triu(T) + np.triu(T, 1).T
        L, dd = _ldlt(T)
        self.d[r] = dd; self.invL[r] = np.linalg.inv(L)
        self.t(r, r)[:] = np.triu(L.T, 1) + np.diag(dd)
        self.diag[r] = True
        self.chain_pos[i] = r + 1
        return True

    def run(self, critical_workers, other_workers):
        """Workers take tickets in list order and block on their task; returns False on a deadlock."""
        tasks = self.pl["tasks"]; n0 = self.pl["n_tasks0"]
        nxt = [0, n0]; end = [n0, len(tasks)]
        workers = [dict(pref=0, task=None) for _ in range(critical_workers)] + [dict(pref=1, task=None) for _ in range(other_workers)]
        while True:
            progress = False
            for w in workers:
                if w["task"] is None:
                    for lst in (w["pref"], 1 - w["pref"]):
                        if nxt[lst] < end[lst]:
                            w["task"] = nxt[lst]; nxt[lst] += 1; progress = True
                            break
            for w in workers:
                if w["task"] is not None and self.task_ready(tasks[w["task"]]):
                    self.run_task(tasks[w["task"]]); w["task"] = None; progress = True
            for i in range(len(self.chain_pos)):
                progress = self.chain_step(i) or progress
            done = nxt[0] == end[0] and nxt[1] == end[1] and all(w["task"] is None for w in workers) and \
                all(self.chain_pos[i] >= int(self.pl["chains"][i][1]) for i in range(len(self.chain_pos)))
            if done:
                return True
            if not progress:
                return False


def _finish_solve(rp: Replay):
    """border update, dense border solve, masked back substitution through the grid rows; x of all factored rows"""
    pl = rp.pl
    Gf, n_fact, n_pad, nbg, nbf = pl["Gf"], pl["n_fact"], pl["n_pad"], pl["nbg"], pl["nbf"]
    Zg = rp.Z[:Gf, :]
    Cb = np.triu(rp.Z[Gf:, Gf:]) - np.triu(Zg[:, Gf:].T @ rp.X[:, Gf:])      # C -= L^T X, upper tiles
    A = Cb[:n_fact - Gf, :n_fact - Gf]
    A = A + np.triu(A, 1).T
    xa = np.linalg.solve(A, Cb[:n_fact - Gf, n_pad - 1 - Gf])
    x = np.zeros(n_fact)
    x[Gf:] = xa
    zg = rp.Z[:Gf, n_pad - 1]                                                # D^-1 L^-1 b of the grid rows
    for r in range(nbg - 1, -1, -1):
        acc = np.zeros(B)
        for c in range(r + 1, nbf):
            if (int(pl["rowmask"][r, c >> 6]) >> (c & 63)) & 1:
                acc += rp.t(r, c) @ x[B * c:B * c + B]
        x[B * r:B * r + B] = rp.invL[r].T @ (zg[B * r:B * r + B] - acc)
    return x


CASES = [
    # cameras, N, P, strips
    ([_cam(CENTRAL_GENERIC, 24, 18)], 4, 12, 1),
    ([_cam(CENTRAL_GENERIC, 24, 18)], 4, 12, 2),
    ([_cam(CENTRAL_GENERIC, 40, 12)], 3, 10, 4),
    ([_cam(CENTRAL_GENERIC, 12, 30)], 3, 10, 3),          # tall grid: the strips cut y
    ([_cam(NONCENTRAL_GENERIC, 16, 10)], 3, 9, 2),
    ([_cam(CENTRAL_GENERIC, 20, 10), _cam(CENTRAL_GENERIC, 26, 12)], 3, 8, 2),      # rig: two independent grids
    ([_cam(CENTRAL_GENERIC, 16, 12)], 2, 5, 0),            # automatic strip count
    ([_cam(CENTRAL_GENERIC, 20, 12)], 16, 40, 2),          # a border of four block columns: two REG2 pairs per row
]


@pytest.mark.parametrize("cams,N,P,strips", CASES)
def test_layout(cams, N, P, strips):
    pl = engine.gridfirst_plan(cams, N, P, strips)
    G = sum(c.params_per_grid_point * c.grid_points for c in cams)
    assert pl["G"] == G and pl["Gf"] % 128 == 0 and pl["n_fact"] % 64 == 0 and pl["n_pad"] % 128 == 0
    assert pl["n_fact"] < pl["n_pad"] and pl["Gf"] + pl["n_border"] <= pl["n_fact"]
    f = pl["f_of_grid"]
    assert np.unique(f).size == G and f.min() >= 0 and f.max() < pl["Gf"]
    assert np.all(np.diff(f) > 0)                        # the engine's grid order IS the order of F (padding only adds gaps)
    for c, cam in enumerate(cams):
        assert sorted(pl["gperm"][c].tolist()) == list(range(cam.grid_points))
    ch = pl["chains"]
    assert ch[0][0] == 0 and ch[-1][1] == pl["nbg"] and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
    # no two tasks write the same tile; every tile of a row mask is produced by exactly one task or by a chain
    seen = set()
    for t in pl["tasks"]:
        kind = int(t[0]) & 255
        for cc in ((int(t[2]), int(t[2]) + 1) if kind == 4 else (int(t[2]),)):
            key = ({3: 1, 4: 2}.get(kind, kind), int(t[1]) if kind in (0, 2, 4) else cc, cc)
            assert key not in seen
            seen.add(key)


@pytest.mark.parametrize("single_tile_tasks", [False, True])
@pytest.mark.parametrize("cams,N,P,strips", CASES)
def test_replay_gives_the_dense_solution(cams, N, P, strips, single_tile_tasks):
    rng = np.random.default_rng(11)
    pl = engine.gridfirst_plan(cams, N, P, strips, single_tile_tasks)
    assert (4 in set(int(k) & 255 for k in pl["tasks"][:, 0])) == (not single_tile_tasks and pl["nbf"] - pl["nbg"] >= 2)
    F, real = _build_system(cams, N, P, pl, rng)
    n_pad, n_fact = pl["n_pad"], pl["n_fact"]
    rp = Replay(pl, F)
    assert rp.run(critical_workers=1, other_workers=1), "the task lists deadlock with one workgroup per list"
    x = _finish_solve(rp)
    Fs = F[:n_fact, :n_fact] + np.triu(F[:n_fact, :n_fact], 1).T
    x_ref = np.linalg.solve(Fs, F[:n_fact, n_pad - 1])
    assert np.max(np.abs(x - x_ref)) <= 1e-9 * np.max(np.abs(x_ref))
    # the factor has nothing outside the tiles the plan knows (the replay never wrote there, the input had nothing there)
    L = np.linalg.cholesky(Fs)
    nbg, nbf = pl["nbg"], pl["nbf"]
    for r in range(nbg):
        for c in range(r + 1, nbf):
            if not (int(pl["rowmask"][r, c >> 6]) >> (c & 63)) & 1:
                assert np.all(L[B * c:B * c + B, B * r:B * r + B] == 0.0)


def _check_plan_by_replay(cams, N, P, strips, single_tile_tasks, seed, workers=(1, 1)):
    pl = engine.gridfirst_plan(cams, N, P, strips, single_tile_tasks)
    F, _ = _build_system(cams, N, P, pl, np.random.default_rng(seed))
    n_pad, n_fact, nbg, nbf = pl["n_pad"], pl["n_fact"], pl["nbg"], pl["nbf"]
    rp = Replay(pl, F)
    assert rp.run(*workers), "the task lists deadlock"
    x = _finish_solve(rp)
    Fs = F[:n_fact, :n_fact] + np.triu(F[:n_fact, :n_fact], 1).T
    x_ref = np.linalg.solve(Fs, F[:n_fact, n_pad - 1])
    assert np.max(np.abs(x - x_ref)) <= 1e-9 * np.max(np.abs(x_ref))
    L = np.linalg.cholesky(Fs)
    for r in range(nbg):
        for c in range(r + 1, nbf):
            if not (int(pl["rowmask"][r, c >> 6]) >> (c & 63)) & 1:
                assert np.all(L[B * c:B * c + B, B * r:B * r + B] == 0.0)


def test_random_geometries_replay_to_the_dense_solution():
    """Property test over the plan's inputs (hypothesis, derandomised: the same 60 geometries on every run): any grid from 4 x 4 up, either
    model, one to three cameras of different sizes, any strip request, both task granularities, one or several simulated workgroups --
    the replay never deadlocks, x is the dense solution, and the factor has nothing outside the planned tiles."""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies

    cam_st = st.tuples(st.sampled_from([CENTRAL_GENERIC, NONCENTRAL_GENERIC]), st.integers(4, 26), st.integers(4, 22))

    @hyp.settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(cams=st.lists(cam_st, min_size=1, max_size=3), N=st.integers(1, 5), P=st.integers(3, 10), strips=st.integers(0, 5),
               single=st.booleans(), seed=st.integers(0, 1000), many=st.booleans())
    def run(cams, N, P, strips, single, seed, many):
        cams = [_cam(m, gw, gh) for m, gw, gh in cams]
        hyp.assume(sum(c.params_per_grid_point * c.grid_points for c in cams) <= 2600)      # keeps one replay around a second
        _check_plan_by_replay(cams, N, P, strips, single, seed, workers=(3, 9) if many else (1, 1))

    run()


def test_more_workers_and_other_seeds_agree():
    cams, N, P = [_cam(CENTRAL_GENERIC, 40, 12)], 3, 10
    pl = engine.gridfirst_plan(cams, N, P, 3)
    F, _ = _build_system(cams, N, P, pl, np.random.default_rng(5))
    xs = []
    for cw, ow in ((1, 1), (2, 7), (5, 40)):
        rp = Replay(pl, F)
        assert rp.run(cw, ow)
        xs.append(_finish_solve(rp))
    assert np.allclose(xs[0], xs[1], rtol=0, atol=1e-12 * np.max(np.abs(xs[0])))
    assert np.allclose(xs[0], xs[2], rtol=0, atol=1e-12 * np.max(np.abs(xs[0])))


def test_two_strips_halve_the_pivot_chain_at_the_headline_configuration_for_free():
    cam = Camera(CENTRAL_GENERIC, 2048, 1456, 0, 0, 2047, 1455, 84, 60)
    one = engine.gridfirst_plan([cam], 500, 815, 1)
    auto = engine.gridfirst_plan([cam], 500, 815, 0)
    assert one["half_bandwidth"] == 367 and one["n_chains"] == 1 and one["chains"][0][1] - one["chains"][0][0] == 158
    longest = lambda p, dep: max([int(c[1] - c[0]) for c in p["chains"] if c[2] == dep] or [0])
    assert auto["strips0"] == 2 and longest(auto, 0) + longest(auto, 1) <= 84
    assert auto["n_border"] == 5445
    # two strips eliminated towards their separator cost no more than one band; every further strip fills a separator
    assert auto["flops"][0] <= 1.02 * one["flops"][0]
    assert engine.gridfirst_plan([cam], 500, 815, 4)["flops"][0] >= 1.3 * one["flops"][0]
