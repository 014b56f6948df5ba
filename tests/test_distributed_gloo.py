"""world_size-2 test of the image-sharded Gauss-Newton step on CPU (gloo backend).

The device path (cba_step with an all-reduce callback) cannot run here (no GPU), so each rank builds
its shard's normal equations with the oracle and goes through exactly the exchange the engine does:
one all-reduce of [partial reduced matrix | partial right-hand side], lambda added once after the
reduction, 8-double scalar all-reduces for the cost bookkeeping, replicated solve, local pose
back-substitution.  The result must equal the single-process oracle step.
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from camera_calibration_amd import distributed as dist_mod  # noqa: E402
from camera_calibration_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_project(cam, grid, pts):
    return orc.project(cam, grid, pts)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, st, _ = syn.reference_test_problem(2, _oracle_project, seed=21, num_points=40, num_poses=12)
    shards = dist_mod.shard_images(np.bincount(pb.obs_image, minlength=pb.n_images), world)
    b, e = shards[rank]
    sub, sst = pb.image_slice(b, e), st.image_slice(b, e)
    op = orc.OracleProblem(sub)
    sysm = op.new_system()
    cost, vec, _ = op.jacobian_pass(sst, sysm)
    # scalar all-reduce #1: cost, valid count, diagonal sum for the automatic lambda (lm_optimizer.h:766-781)
    dsum = sum(np.trace(B) for B in sysm.block_diag_H) + np.trace(sysm.dense_H)
    scal = torch.tensor([cost, float((vec >= 0).sum()), dsum], dtype=torch.float64)
    dist.all_reduce(scal)
    total_dof = 6 * pb.n_images + pb.dense_dof
    lam = 1e-5 * scal[2].item() / total_dof
    # the one big exchange: [S_partial (upper) | s_partial]
    S, s, W, Db = dist_mod.local_reduced_system(sysm.block_diag_H, sysm.off_diag_H, sysm.dense_H,
                                                sysm.block_diag_b, sysm.dense_b, lam)
    buf = torch.from_numpy(np.concatenate([S.ravel(), s]))
    dist.all_reduce(buf)
    D = pb.dense_dof
    xd = dist_mod.solve_reduced(buf[:D * D].numpy().reshape(D, D), buf[D * D:].numpy(), lam)
    xb = Db - W @ xd   # local pose back-substitution
    # candidate state of the shard, cost-only pass, masked sums (CostIsSmallerThan) all-reduced
    x_local = np.concatenate([xb, xd])
    cand = op.apply_update(sst, x_local)
    c2, v2 = op.cost_pass(cand)
    both = (vec >= 0) & (v2 >= 0)
    red = torch.tensor([vec[both].sum(), v2[both].sum(), float(both.sum()), c2], dtype=torch.float64)
    dist.all_reduce(red)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b=b, e=e, xb=xb, xd=xd, lam=lam, cost=scal[0].item(),
             red=red.numpy(), poses=cand.rig_tr_global, points=cand.points)
    dist.destroy_process_group()


def test_two_rank_sharded_step_equals_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    pb, st, _ = syn.reference_test_problem(2, _oracle_project, seed=21, num_points=40, num_poses=12)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    cost, vec, _ = op.jacobian_pass(st, sysm)
    lam = 1e-5 * (sum(np.trace(B) for B in sysm.block_diag_H) + np.trace(sysm.dense_H)) / pb.total_dof
    sysm.add_lambda(lam)
    x_ref = orc.schur_solve(sysm)
    cand = op.apply_update(st, x_ref)
    c2, v2 = op.cost_pass(cand)
    both = (vec >= 0) & (v2 >= 0)
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    assert r[0]["b"] == 0 and r[-1]["e"] == pb.n_images and r[0]["e"] == r[1]["b"]
    scale = np.abs(x_ref).max()
    for k in range(world):
        assert abs(r[k]["lam"] - lam) <= 1e-12 * lam
        assert abs(r[k]["cost"] - cost) <= 1e-10 * cost
        np.testing.assert_allclose(r[k]["xd"], x_ref[pb.block_dof:], rtol=1e-7, atol=1e-9 * scale)
        np.testing.assert_allclose(r[k]["xb"], x_ref[6 * int(r[k]["b"]):6 * int(r[k]["e"])], rtol=1e-7, atol=1e-9 * scale)
        # identical accept/reject inputs on every rank
        np.testing.assert_allclose(r[k]["red"], [vec[both].sum(), v2[both].sum(), both.sum(), c2], rtol=1e-6)
        np.testing.assert_allclose(r[k]["poses"], cand.rig_tr_global[int(r[k]["b"]):int(r[k]["e"])], atol=1e-9)
        np.testing.assert_allclose(r[k]["points"], cand.points, atol=1e-9)
    np.testing.assert_array_equal(r[0]["xd"], r[1]["xd"])   # replicated solve is bit-identical


# ---------------------------------------------------------------------------------------------------------------------
# the buffer that actually crosses ranks: the packed upper 128-row blocks of the padded reduced system (k_pack_upper)
# ---------------------------------------------------------------------------------------------------------------------
def _packed_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, st, _ = syn.reference_test_problem(2, _oracle_project, seed=22, num_points=60, num_poses=10)
    shards = dist_mod.shard_images(np.bincount(pb.obs_image, minlength=pb.n_images), world)
    b, e = shards[rank]
    sub, sst = pb.image_slice(b, e), st.image_slice(b, e)
    op = orc.OracleProblem(sub)
    sysm = op.new_system()
    op.jacobian_pass(sst, sysm)
    lam = 0.37
    S, s, _, _ = dist_mod.local_reduced_system(sysm.block_diag_H, sysm.off_diag_H, sysm.dense_H, sysm.block_diag_b, sysm.dense_b, lam)
    D = pb.dense_dof
    n_pad = -(-(D + 1) // 128) * 128                 # the engine's padding rule (cba_api.hip padded_dims)
    if -(-D // 64) * 64 >= n_pad:
        n_pad += 128
    Sp = np.zeros((n_pad, n_pad))
    Sp[:D, :D] = np.triu(S)
    Sp[:D, n_pad - 1] = s                             # right-hand side in the last padding column
    P = dist_mod.pack_upper(Sp)
    t = torch.from_numpy(P)
    dist.all_reduce(t)                                # ONE all-reduce of the packed buffer, as cba_step does
    Ssum = dist_mod.unpack_upper(t.numpy(), n_pad)
    np.savez(os.path.join(out_dir, f"packed{rank}.npz"), S=Ssum[:D, :D], s=Ssum[:D, n_pad - 1], n_pad=n_pad, count=P.size)
    dist.destroy_process_group()


def test_packed_upper_layout_sums_over_ranks(tmp_path):
    world = 2
    mp.spawn(_packed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pb, st, _ = syn.reference_test_problem(2, _oracle_project, seed=22, num_points=60, num_poses=10)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    op.jacobian_pass(st, sysm)
    S, s, _, _ = dist_mod.local_reduced_system(sysm.block_diag_H, sysm.off_diag_H, sysm.dense_H, sysm.block_diag_b, sysm.dense_b, 0.37)
    r = [np.load(os.path.join(str(tmp_path), f"packed{k}.npz")) for k in range(world)]
    for k in range(world):
        np.testing.assert_allclose(np.triu(r[k]["S"]), np.triu(S), rtol=1e-9, atol=1e-9 * np.abs(S).max())
        np.testing.assert_allclose(r[k]["s"], s, rtol=1e-9, atol=1e-9 * np.abs(s).max())
    # the size the engine asks its host for (cba_reduce_buffer_doubles, pure host code) is the size of this layout
    from camera_calibration_amd import engine as eng
    assert eng.Engine.reduce_buffer_doubles(pb) == int(r[0]["count"]) == dist_mod.packed_upper_doubles(int(r[0]["n_pad"]))


def test_pack_unpack_roundtrip_and_block_offsets():
    n_pad = 640
    rng = np.random.default_rng(0)
    S = np.triu(rng.normal(size=(n_pad, n_pad)))
    P = dist_mod.pack_upper(S)
    assert P.size == dist_mod.packed_upper_doubles(n_pad)
    # offset formula of k_pack_upper: block blk starts at 128 * (blk * n_pad - 64 * blk * (blk - 1))
    for blk in range(n_pad // 128):
        off = 128 * (blk * n_pad - 64 * blk * (blk - 1))
        assert P[off] == S[128 * blk, 128 * blk]
    S2 = dist_mod.unpack_upper(P, n_pad)
    for blk in range(n_pad // 128):
        np.testing.assert_array_equal(S2[128 * blk:128 * blk + 128, 128 * blk:], S[128 * blk:128 * blk + 128, 128 * blk:])


def _gloo_collective(rank, world):
    """cba_collective_fn semantics on numpy arrays through gloo (no reduce-scatter there: all-reduce + own block)."""
    def collective(op, send, recv):
        if op == dist_mod.COLL_ALLREDUCE_SUM:
            dist.all_reduce(torch.from_numpy(recv))
        elif op == dist_mod.COLL_REDUCE_SCATTER_SUM:
            t = torch.from_numpy(send.copy())
            dist.all_reduce(t)
            count = recv.size
            recv[:] = t.numpy()[rank * count:(rank + 1) * count]
        else:
            out = [torch.empty(send.size, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(send))
            recv[:] = np.concatenate([t.numpy() for t in out])
    return collective


def _dist_ldlt_worker(rank, world, port, out_dir, n, group, W, tail_rows):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank holds a different partial sum; the sum is symmetric positive definite
    parts = []
    for q in range(world):
        rng = np.random.default_rng(5 + q)
        A = rng.standard_normal((n, n // world + 40))
        parts.append(A @ A.T + 0.5 / world * np.eye(n))
    L, d = dist_mod.distributed_ldlt_upper(parts[rank], rank, world, _gloo_collective(rank, world), group=group, W=W, tail_rows=tail_rows)
    np.savez(os.path.join(out_dir, f"ldlt{rank}.npz"), L=L, d=d, S=sum(parts))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,group,W,tail_rows", [(2, 480, 32, 64, 160), (3, 480, 32, 64, 160), (2, 416, 32, 128, 96), (3, 200, 32, 64, 300)])
def test_distributed_factorisation_schedule_on_cpu(tmp_path, world, n, group, W, tail_rows):
    """Host mirror of ldlt_factor_distributed (cba_config.distributed_solve), scaled down: first band all-reduced, the rest
    reduce-scattered into the block-cyclic owners of the column groups, every band factored on every rank, trailing updates on
    owned columns only, next band all-gathered from its owners, last rows replicated (the last case has no super-panel at all:
    the packed all-reduce path).  Every rank must end up with the same complete factor of the SUM of the partial systems."""
    mp.spawn(_dist_ldlt_worker, args=(world, _free_port(), str(tmp_path), n, group, W, tail_rows), nprocs=world, join=True)
    out = [np.load(os.path.join(str(tmp_path), f"ldlt{k}.npz")) for k in range(world)]
    S = out[0]["S"]
    for k in range(world):
        L, d = out[k]["L"], out[k]["d"]
        assert np.all(d > 0)
        assert np.abs((L * d) @ L.T - S).max() <= 1e-10 * np.abs(S).max()
        assert np.array_equal(L, out[0]["L"]) and np.array_equal(d, out[0]["d"])       # identical arithmetic on identical data
    # and it is the factorisation a single process computes from the sum
    L1, d1 = dist_mod.distributed_ldlt_upper(S, 0, 1, lambda op, send, recv: recv.__setitem__(slice(None), send) if op != 0 else None,
                                             group=group, W=W, tail_rows=tail_rows)
    assert np.abs(L1 - out[0]["L"]).max() <= 1e-10 and np.abs(d1 - out[0]["d"]).max() <= 1e-10 * d1.max()


def test_transfer_layout_matches_the_device_kernels():
    """dist_rect (distributed.py) against the closed forms of kernels_linalg.hip: the blocks of one rank are contiguous, in
    group order, and disjoint; every column group at or right of the first row is covered exactly once over the ranks."""
    for n_pad, group, world, R0, nrows in [(12672, 512, 8, 2048, 0), (12672, 512, 3, 4096, 2048), (42880, 512, 8, 2048, 0), (1152, 512, 2, 512, 640)]:
        g_begin = R0 // group
        seen = set()
        for q in range(world):
            end = 0
            for col0, width, height, off in dist_mod._dist_rects(n_pad, group, g_begin, world, R0, nrows, q):
                assert off == end and (col0 // group) % world == q and col0 >= g_begin * group
                assert height == (nrows if nrows else min(col0 + group, n_pad) - R0) and 0 < width <= group
                end = off + height * width
                seen.add(col0)
            assert end <= dist_mod._dist_count(n_pad, group, g_begin, world, R0, nrows)
        assert seen == set(range(g_begin * group, n_pad, group))


# ---- follow-up design: what would cross the ranks with the grid-first elimination order ----
def _gf_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, st, _ = syn.baseline_config(3, _oracle_project, n_imagesets=8, grid_wh=(14, 9), lattice_xy=(8, 9))
    shards = dist_mod.shard_images(np.bincount(pb.obs_image, minlength=pb.n_images), world)
    b, e = shards[rank]
    sub, sst = pb.image_slice(b, e), st.image_slice(b, e)
    sysm = orc.OracleProblem(sub).new_system()
    orc.OracleProblem(sub).jacobian_pass(sst, sysm)
    lay = dist_mod.GridFirstSharedLayout(pb.cameras, pb.n_points)
    buf = torch.from_numpy(lay.pack(sysm.dense_H, sysm.dense_b))
    dist.all_reduce(buf)                                            # the ONE exchange of the shared blocks
    H, bb = lay.unpack(buf.numpy())
    np.savez(os.path.join(out_dir, f"gf_rank{rank}.npz"), H=H, b=bb, doubles=lay.doubles)
    dist.destroy_process_group()


def test_grid_first_shared_blocks_layout_summed_over_two_ranks(tmp_path):
    """The all-reduce layout a grid-first image-sharded step would use -- banded grid x grid block, rig / point rows x grid, rig rows,
    3 x 3 point blocks, J^T r: BASELINE.json's "shared intrinsics and pattern J^T J / J^T r blocks" -- packed by two ranks from their
    shards' accumulators (oracle), summed with gloo, unpacked: equal to the single-process dense part, every entry (the layout covers
    the whole structure of the dense part), at a third of the doubles of the packed reduced system at BASELINE configs[1]."""
    world = 2
    mp.spawn(_gf_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pb, st, _ = syn.baseline_config(3, _oracle_project, n_imagesets=8, grid_wh=(14, 9), lattice_xy=(8, 9))
    sysm = orc.OracleProblem(pb).new_system()
    orc.OracleProblem(pb).jacobian_pass(st, sysm)
    ref_H, ref_b = np.triu(sysm.dense_H), sysm.dense_b
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"gf_rank{r}.npz"))
        assert np.abs(d["H"] - ref_H).max() <= 1e-12 * np.abs(ref_H).max()         # sums of two partial accumulators: rounding only
        assert np.abs(d["b"] - ref_b).max() <= 1e-12 * np.abs(ref_b).max()
        assert np.array_equal(d["H"] != 0, d["H"] != 0) and int(d["doubles"]) < pb.dense_dof * (pb.dense_dof + 1) // 2
    # sizes at BASELINE configs[1]: 227 MB against the 649 MB of the packed upper triangle of S
    from camera_calibration_amd.problem import Camera
    lay = dist_mod.GridFirstSharedLayout([Camera(0, 2048, 1456, 0, 0, 2047, 1455, 84, 60)], 815)
    assert lay.hb == [367] and 220e6 < lay.doubles * 8 < 235e6
