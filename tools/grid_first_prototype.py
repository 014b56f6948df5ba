"""CPU prototype of the grid-first (bordered block-sparse) elimination -- DEVELOPER TOOL, not shipped, not a test.

What it checks, on the oracle's own normal equations of a BASELINE configs[1]-shaped problem (one central-generic
camera, planar pattern, perturbed start):

  1. STRUCTURE.  With the grid unknowns ordered along the short grid dimension the grid x grid block of J^T J is banded:
     an observation touches a 4 x 4 window of control points (APP/models/central_grid.h:199-209), so two unknowns couple
     only if their control points are at most 3 apart in both directions: half-bandwidth (3 gh + 3) ppg + ppg - 1.
     With `--strips S` the grid columns are cut into S vertical strips separated by 3-column separators that are
     eliminated after all strips (the order the engine uses: independent pivot chains); the script checks that the factor
     has NO entry outside the block pattern the host-side symbolic analysis predicts.
  2. NUMERICS.  x from (a) a dense solve refined in extended precision (the reference answer), (b) the pose-first Schur
     order of the reference (LV/lm_optimizer.h:1247-1369: eliminate the 6 x 6 pose blocks, factor the dense D x D rest),
     (c) the grid-first order (eliminate the grid by a banded LDL^T, factor the (6 N + 3 P) border) -- for a range of
     lambda.  SURVEY fact 3: (H + lambda I) x = b has one solution, any exact elimination order may be used.
  3. FLOPS of the two orders at the problem's size and at BASELINE configs[1] / [2] / [3].

Usage:  python tools/grid_first_prototype.py [--grid 24x18] [--images 16] [--strips 1]
The oracle (oracle/) is the checker here, as in tests/; nothing in the product imports this file.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from camera_calibration_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def full_system(op, st):
    """H (upper + lower filled), b in the reference's variable order: [poses 6N | points 3P | grid]."""
    sysm = op.new_system()
    op.jacobian_pass(st, sysm)
    nb, bs, dd = sysm.n_blocks, sysm.block_size, sysm.dense_dof
    n = nb * bs + dd
    H = np.zeros((n, n))
    for i in range(nb):
        blk = np.triu(sysm.block_diag_H[i])
        H[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] = blk + np.triu(blk, 1).T
    H[:nb * bs, nb * bs:] = sysm.off_diag_H
    H[nb * bs:, :nb * bs] = sysm.off_diag_H.T
    D = np.triu(sysm.dense_H)
    H[nb * bs:, nb * bs:] = D + np.triu(D, 1).T
    b = np.concatenate([sysm.block_diag_b, sysm.dense_b])
    return H, b, nb * bs, dd


def grid_order(gw, gh, ppg, strips):
    """Elimination order of the control points (list of (gx, gy)): along the SHORT dimension inside a column of the long
    one; `strips` vertical strips first, their 3-column separators last.  Returns (order, group) with group[i] = strip
    index (0 .. S-1) or S + separator index."""
    long_is_x = gw >= gh
    nl, ns = (gw, gh) if long_is_x else (gh, gw)
    S = max(1, strips)
    # separators: 3 lines each, evenly spaced
    seps = []
    if S > 1:
        interior = nl - 3 * (S - 1)
        base, extra = divmod(interior, S)
        pos = 0
        for s in range(S - 1):
            pos += base + (1 if s < extra else 0)
            seps.append((pos, pos + 3))
            pos += 3
    order, group = [], []
    bounds = [0] + [e for (_, e) in seps]
    ends = [b for (b, _) in seps] + [nl]
    for s in range(S):
        for l in range(bounds[s], ends[s]):
            for t in range(ns):
                order.append((l, t) if long_is_x else (t, l)); group.append(s)
    for k, (b0, b1) in enumerate(seps):
        for l in range(b0, b1):
            for t in range(ns):
                order.append((l, t) if long_is_x else (t, l)); group.append(S + k)
    return order, np.array(group)


def chol_like_ldlt(A):
    """unpivoted LDL^T via numpy's Cholesky of the (positive definite) matrix: L_c = L sqrt(D)"""
    C = np.linalg.cholesky(A)
    dsq = np.diag(C).copy()
    return C / dsq, dsq * dsq


def solve_ldlt(L, d, b):
    import scipy.linalg as sl
    y = sl.solve_triangular(L, b, lower=True, unit_diagonal=True)
    return sl.solve_triangular(L.T, y / d, lower=False, unit_diagonal=True)


def refine(H, b, x, iters=4):
    Hq = H.astype(np.longdouble); bq = b.astype(np.longdouble); xq = x.astype(np.longdouble)
    L, d = chol_like_ldlt(H)
    for _ in range(iters):
        r = (bq - Hq @ xq).astype(np.float64)
        xq = xq + solve_ldlt(L, d, r).astype(np.longdouble)
    return xq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="24x18")
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--strips", type=int, default=1)
    ap.add_argument("--lattice", default="12x17")
    a = ap.parse_args()
    gw, gh = (int(v) for v in a.grid.split("x"))
    lx, ly = (int(v) for v in a.lattice.split("x"))
    pb, st, _ = synthetic.baseline_config(2, lambda c, g, p: orc.project(c, g, p), n_imagesets=a.images, grid_wh=(gw, gh),
                                          lattice_xy=(lx, ly))
    op = orc.OracleProblem(pb)
    H, b, nblk, dd = full_system(op, st)
    n = H.shape[0]
    N, P, ppg = pb.n_images, pb.n_points, 2
    G = ppg * gw * gh
    g0 = 6 * N + 3 * P                      # first grid unknown in the reference order (row-major grid: gx + gy gw)
    print(f"problem: N={N} P={P} grid {gw}x{gh}  unknowns {n} = poses {6 * N} + points {3 * P} + grid {G}; obs {pb.n_obs}")

    # ---- 1. structure ----
    order, group = grid_order(gw, gh, ppg, a.strips)
    perm_g = np.array([g0 + ppg * (gx + gy * gw) + dpar for (gx, gy) in order for dpar in range(ppg)])
    grp_u = np.repeat(group, ppg)
    Hgg = H[np.ix_(perm_g, perm_g)]
    nzr, nzc = np.nonzero(Hgg)
    bw = int(np.max(np.abs(nzr - nzc))) if a.strips <= 1 else None
    short = min(gw, gh)
    print(f"grid x grid block: measured half-bandwidth {bw}, predicted (3*{short}+3)*{ppg}+{ppg - 1} = {(3 * short + 3) * ppg + ppg - 1}"
          if bw is not None else f"grid x grid block ordered as {a.strips} strips + {a.strips - 1} separators")
    # border = [points | poses sorted by the first grid unknown they touch]
    Hga = H[np.ix_(perm_g, np.arange(0, g0))]
    first = np.array([np.argmax(np.abs(Hga[:, 6 * i:6 * i + 6]).sum(axis=1) > 0) for i in range(N)])
    pose_order = np.argsort(first, kind="stable")
    border = np.concatenate([np.arange(6 * N, 6 * N + 3 * P)] + [np.arange(6 * i, 6 * i + 6) for i in pose_order])
    perm = np.concatenate([perm_g, border])
    A = H[np.ix_(perm, perm)]
    bp = b[perm]

    # symbolic block pattern (64 x 64 blocks would be the engine's; here: per UNKNOWN, exact) -> predicted fill
    lam_struct = 1e-3 * np.mean(np.diag(H))
    L, d = chol_like_ldlt(A + lam_struct * np.eye(n))
    pat = (A != 0)
    # boolean elimination on the strip / separator / border GROUPS (what the host analysis does per 64-block)
    ngrp = int(grp_u.max()) + 2
    gid = np.concatenate([grp_u, np.full(n - G, ngrp - 1)])
    Bpat = np.zeros((ngrp, ngrp), bool)
    for gi in range(ngrp):
        for gj in range(ngrp):
            Bpat[gi, gj] = pat[np.ix_(gid == gi, gid == gj)].any()
    for k in range(ngrp):
        for i in range(k + 1, ngrp):
            if Bpat[i, k]:
                for j in range(k + 1, ngrp):
                    if Bpat[j, k]:
                        Bpat[i, j] = Bpat[j, i] = True
    viol = 0.0
    for gi in range(ngrp):
        for gj in range(gi):
            if not Bpat[gi, gj]:
                viol = max(viol, np.abs(L[np.ix_(gid == gi, gid == gj)]).max(initial=0.0))
    print(f"largest factor entry outside the predicted group pattern: {viol:.1e} (must be 0)")
    # inside a strip: band
    if a.strips > 1:
        hb = (3 * short + 3) * ppg + ppg - 1
        worst = 0.0
        for s in range(a.strips):
            idx = np.nonzero(grp_u == s)[0]
            Ls = L[np.ix_(idx, idx)]
            r, c = np.nonzero(Ls)
            worst = max(worst, float(np.max(r - c)))
        print(f"largest band offset of the factor inside a strip: {worst:.0f} (predicted <= {hb})")

    # ---- 2. numerics ----
    print(f"{'lambda/mean diag':>18} {'direct':>10} {'pose-first':>11} {'grid-first':>11}")
    md = np.mean(np.diag(H))
    import scipy.linalg as sl
    for rel in (1e-3, 1e-5, 4e-8, 1e-9):
        lam = rel * md
        Hl = H + lam * np.eye(n)
        x_direct = np.linalg.solve(Hl, b)
        x_ref = refine(Hl, b, x_direct)
        # pose-first: Schur on the 6 x 6 pose blocks
        Dinv = np.zeros((nblk, nblk))
        for i in range(N):
            Dinv[6 * i:6 * i + 6, 6 * i:6 * i + 6] = np.linalg.inv(Hl[6 * i:6 * i + 6, 6 * i:6 * i + 6])
        Bm = Hl[:nblk, nblk:]
        Sm = Hl[nblk:, nblk:] - Bm.T @ (Dinv @ Bm)
        sv = b[nblk:] - Bm.T @ (Dinv @ b[:nblk])
        Ls, ds = chol_like_ldlt(Sm)
        xd = solve_ldlt(Ls, ds, sv)
        xb = Dinv @ (b[:nblk] - Bm @ xd)
        x_pose = np.concatenate([xb, xd])
        # grid-first: banded LDL^T of the grid block, border Schur complement
        Al = A + lam * np.eye(n)
        Lg, dg = chol_like_ldlt(Al[:G, :G])
        Y = sl.solve_triangular(Lg, Al[:G, G:], lower=True, unit_diagonal=True)
        yb = sl.solve_triangular(Lg, bp[:G], lower=True, unit_diagonal=True)
        Wm = Y / dg[:, None]
        Sa = Al[G:, G:] - Y.T @ Wm
        sa = bp[G:] - Wm.T @ yb
        La, da = chol_like_ldlt(Sa)
        xa = solve_ldlt(La, da, sa)
        xg = sl.solve_triangular(Lg.T, yb / dg - Wm @ xa, lower=False, unit_diagonal=True)
        x_grid = np.zeros(n)
        x_grid[perm] = np.concatenate([xg, xa])
        nrm = float(np.max(np.abs(x_ref)))
        err = lambda x: float(np.max(np.abs(x.astype(np.longdouble) - x_ref))) / nrm
        print(f"{rel:18.0e} {err(x_direct):10.1e} {err(x_pose):11.1e} {err(x_grid):11.1e}")

    # ---- 3. flops ----
    def flops(N, P, C, gw, gh, ppg):
        G = C * ppg * gw * gh
        Abd = 6 * N + 3 * P + (6 * C if C > 1 else 0)
        D = G + 3 * P + (6 * C if C > 1 else 0)
        hb = (3 * min(gw, gh) + 3) * ppg + ppg - 1
        pose_first = D * D * 6 * N + D ** 3 / 3.0          # dense Schur product (upper) + factorisation
        grid_first = G * hb * hb + 2.0 * G * hb * Abd + Abd * Abd * G + Abd ** 3 / 3.0
        return pose_first, grid_first, D, Abd
    for name, args in (("this problem", (N, P, 1, gw, gh, 2)), ("configs[1]", (500, 815, 1, 84, 60, 2)),
                       ("configs[2]", (1000, 815, 2, 84, 60, 2)), ("configs[3]", (800, 815, 1, 52, 40, 5)),
                       ("configs[4]", (4000, 815, 4, 84, 60, 2))):
        pf, gf, D, Ab = flops(*args)
        print(f"{name:>13}: dense D = {D:6d}, border = {Ab:6d}; pose-first {pf / 1e9:9.1f} GFLOP (dense product), grid-first {gf / 1e9:9.1f} GFLOP")


if __name__ == "__main__":
    main()
