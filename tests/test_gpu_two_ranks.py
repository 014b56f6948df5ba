"""The image-sharded path with REAL sharding on one GPU: two processes share device 0, each owns half of the imagesets,
and every Gauss-Newton step goes through cba_step's multi-rank control flow (packed-upper all-reduce of the reduced
system, 8-double scalar reductions, cross-rank failure flag, lambda added after the reduction, replicated factorisation,
local pose back-substitution).  RCCL refuses two ranks on one device, so the reductions are staged through host memory
with gloo (camera_calibration_amd.distributed.make_allreduce_host_staged); the engine code under test is identical to
the RCCL configuration.  Reference for the result: the single-process engine on the whole problem."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from camera_calibration_amd import distributed as dist_mod  # noqa: E402
from camera_calibration_amd import engine as eng  # noqa: E402
from camera_calibration_amd import synthetic as syn  # noqa: E402
from parity_record import check, check_equal  # noqa: E402

pytestmark = pytest.mark.gpu
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(grid_wh=(20, 16)):
    return syn.baseline_config(3, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=24, grid_wh=grid_wh)


DIST_GRID = (30, 24)       # D = 5337: with 512 rows left to the final launch the distributed schedule has two super-panels
DIST_TAIL_ROWS = 512


def _problem_full_grid():
    """BASELINE configs[1] with its full 84 x 60 grid (D = 12 525: super-panels of ~2048 rows and the ~9000-row final launch
    at the DEFAULT schedule parameters) and 60 imagesets."""
    return syn.baseline_config(2, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=60)


def _worker(rank, world, port, out_dir, distributed_solve=False, use_collective=False, full_grid=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng.prepare(0)
    tail_rows = DIST_TAIL_ROWS if (distributed_solve and not full_grid) else 0       # cba_solver_options, per problem
    pb, st, _ = _problem_full_grid() if full_grid else _problem(DIST_GRID if distributed_solve else (20, 16))
    shards = dist_mod.shard_images(np.bincount(pb.obs_image, minlength=pb.n_images), world)
    b, e = shards[rank]
    sub, sst = pb.image_slice(b, e), st.image_slice(b, e)
    allreduce = dist_mod.make_allreduce_host_staged()
    en = eng.Engine(sub, device=0, allreduce=allreduce, n_images_global=pb.n_images, deterministic=True,
                    last_projection=sub.obs_xy.astype(np.float64), distributed_solve=distributed_solve, rank=rank, world_size=world,
                    collective=dist_mod.make_collective_host_staged() if use_collective else None, factor_tail_rows=tail_rows)
    en.set_state(sst)
    lam = -1.0
    reps = []
    for _ in range(STEPS):
        r = en.step(lam)
        lam = r.final_lambda
        reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
    out = en.get_state(sst)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b=b, e=e, reps=np.array(reps), poses=out.rig_tr_global, points=out.points,
             camrig=out.camera_tr_rig, grid0=out.grids[0], grid1=out.grids[-1])
    en.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_process_engine(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pb, st, _ = _problem()
    en = eng.Engine(pb, deterministic=True, last_projection=pb.obs_xy.astype(np.float64))
    en.set_state(st)
    lam = -1.0
    reps = []
    for _ in range(STEPS):
        r = en.step(lam)
        lam = r.final_lambda
        reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
    ref = en.get_state(st)
    en.close()
    reps = np.array(reps)
    case = "2 ranks on 1 GPU (cfg-3-shaped, 24 imagesets) vs single process"
    rk = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    assert rk[0]["b"] == 0 and rk[1]["e"] == pb.n_images and rk[0]["e"] == rk[1]["b"]
    for k in range(world):
        check_equal(case, f"rank {k}: LM attempts / accept decisions / valid counts",
                    int(np.count_nonzero(rk[k]["reps"][:, 3:] != reps[:, 3:])))
        # the first cost pass sees the same state on both sides (only the order of the cross-rank sum differs); everything
        # after the first solve carries the rounding of the partial Schur products amplified by the condition of the reduced
        # system (observed 1.1e-7 on the cost after three iterations)
        check(case, f"rank {k}: initial cost of the first step rel", abs(rk[k]["reps"][0, 0] - reps[0, 0]) / reps[0, 0], 5e-15)
        check(case, f"rank {k}: costs rel", (np.abs(rk[k]["reps"][:, :2] - reps[:, :2]) / reps[:, :2]).max(), 5e-7)     # observed 1.1e-7 (3 iterations)
        check(case, f"rank {k}: lambda rel", (np.abs(rk[k]["reps"][:, 2] - reps[:, 2]) / reps[:, 2]).max(), 5e-13)
        b, e = int(rk[k]["b"]), int(rk[k]["e"])
        check(case, f"rank {k}: own poses abs", np.abs(rk[k]["poses"] - ref.rig_tr_global[b:e]).max(), 5e-8)     # observed 2e-9
        check(case, f"rank {k}: points abs", np.abs(rk[k]["points"] - ref.points).max(), 5e-8)
        check(case, f"rank {k}: camera_tr_rig abs", np.abs(rk[k]["camrig"] - ref.camera_tr_rig).max(), 5e-8)
        check(case, f"rank {k}: grids abs", max(np.abs(rk[k]["grid0"] - ref.grids[0]).max(), np.abs(rk[k]["grid1"] - ref.grids[1]).max()), 1e-7)
    # the replicated part of the state is bit-identical on both ranks (same reduced system, same factorisation)
    for key in ("points", "camrig", "grid0", "grid1"):
        check_equal(case, f"replicated state identical on both ranks: {key}", int(np.count_nonzero(rk[0][key] != rk[1][key])))
    check_equal(case, "step reports identical on both ranks", int(np.count_nonzero(rk[0]["reps"] != rk[1]["reps"])))


@pytest.mark.parametrize("world,use_collective", [(2, False), (3, True)])
def test_two_ranks_with_the_distributed_factorisation(tmp_path, world, use_collective):
    """cba_config.distributed_solve (kernels_linalg.hip: ldlt_factor_distributed): the partial reduced systems are
    reduce-scattered into the block-cyclic owners of the 512-column groups (first band all-reduced), every rank runs the dataflow
    launch of each super-panel on the complete row band, applies the K = 2048 update to its own column groups only (next band
    first) and the next band is all-gathered from its owners; two super-panels at this size (D = 5337, 512 rows left to the
    final launch).  With two ranks the collectives are emulated through the all-reduce callback, with three (uneven image
    shards, three-way ownership) they go through a cba_collective_fn (host-staged gloo).  Same three LM iterations against
    the single-process engine: same decisions, results equal to the rounding of two different summation orders."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, use_collective), nprocs=world, join=True)
    pb, st, _ = _problem(DIST_GRID)
    en = eng.Engine(pb, deterministic=True, last_projection=pb.obs_xy.astype(np.float64), factor_tail_rows=DIST_TAIL_ROWS)
    try:
        en.set_state(st)
        lam = -1.0
        reps = []
        for _ in range(STEPS):
            r = en.step(lam)
            lam = r.final_lambda
            reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
        ref = en.get_state(st)
    finally:
        en.close()
    reps = np.array(reps)
    case = f"{world} ranks on 1 GPU, distributed factorisation (cfg-3-shaped, 24 imagesets, 30x24 grids) vs single process"
    rk = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    for k in range(world):
        check_equal(case, f"rank {k}: LM attempts / accept decisions / valid counts",
                    int(np.count_nonzero(rk[k]["reps"][:, 3:] != reps[:, 3:])))
        check(case, f"rank {k}: initial cost of the first step rel", abs(rk[k]["reps"][0, 0] - reps[0, 0]) / reps[0, 0], 5e-15)
        check(case, f"rank {k}: costs rel", (np.abs(rk[k]["reps"][:, :2] - reps[:, :2]) / reps[:, :2]).max(), 5e-7)     # observed 1.1e-7 (3 iterations)
        b, e = int(rk[k]["b"]), int(rk[k]["e"])
        check(case, f"rank {k}: own poses abs", np.abs(rk[k]["poses"] - ref.rig_tr_global[b:e]).max(), 5e-8)     # observed 2e-9
        check(case, f"rank {k}: points abs", np.abs(rk[k]["points"] - ref.points).max(), 5e-8)
        check(case, f"rank {k}: grids abs", max(np.abs(rk[k]["grid0"] - ref.grids[0]).max(), np.abs(rk[k]["grid1"] - ref.grids[1]).max()), 1e-7)
    for key in ("points", "camrig", "grid0", "grid1"):
        for k in range(1, world):
            check_equal(case, f"replicated state identical on ranks 0 and {k}: {key}", int(np.count_nonzero(rk[0][key] != rk[k][key])))


def test_native_rccl_callback_world_of_one(tmp_path):
    """libcalib_ba_rccl.so (include/cba_rccl.h), the callback a C++ host passes to cba_config.allreduce: a communicator of one
    rank on this GPU must leave every reduction unchanged, i.e. reproduce the plain path."""
    pb, st, _ = syn.baseline_config(2, lambda cam, grid, pts: eng.project(cam, grid, pts), n_imagesets=20, grid_wh=(24, 18))
    rc = dist_mod.NativeRccl(0, 1, str(tmp_path / "rccl_id"), 0)
    e1 = eng.Engine(pb, deterministic=True)
    e2 = eng.Engine(pb, deterministic=True, allreduce_native=(rc.fn, rc.user), n_images_global=pb.n_images)
    e1.set_state(st); e2.set_state(st)
    l1 = l2 = -1.0
    for _ in range(3):
        r1 = e1.step(l1); r2 = e2.step(l2)
        l1, l2 = r1.final_lambda, r2.final_lambda
        assert r1.accepted == r2.accepted and r1.lm_attempts == r2.lm_attempts
        check("native RCCL callback, world 1", "final cost rel", abs(r1.final_cost - r2.final_cost) / r1.final_cost, 5e-8)   # lambda enters before / after the Schur product
    s1, s2 = e1.get_state(st), e2.get_state(st)
    check("native RCCL callback, world 1", "points abs", np.abs(s1.points - s2.points).max(), 5e-8)
    e1.close(); e2.close(); rc.close()


def test_native_rccl_collective_world_of_one_distributed_solve(tmp_path):
    """cba_rccl_collective (ncclAllReduce / ncclReduceScatter / ncclAllGather) behind cba_config.collective with a communicator
    of one rank: the whole distributed schedule (two super-panels, every pack / collective / unpack) must reproduce the
    single-GPU factorisation of the same system -- all columns owned, every transfer a copy through RCCL."""
    pb, st, _ = _problem(DIST_GRID)
    rc = dist_mod.NativeRccl(0, 1, str(tmp_path / "rccl_id"), 0)
    try:
        e1 = eng.Engine(pb, deterministic=True, factor_tail_rows=DIST_TAIL_ROWS)
        e2 = eng.Engine(pb, deterministic=True, allreduce_native=(rc.fn, rc.user), n_images_global=pb.n_images, distributed_solve=True,
                        rank=0, world_size=1, collective_native=(rc.collective_fn, rc.user), factor_tail_rows=DIST_TAIL_ROWS)
        e1.set_state(st); e2.set_state(st)
        l1 = l2 = -1.0
        for _ in range(3):
            r1 = e1.step(l1); r2 = e2.step(l2)
            l1, l2 = r1.final_lambda, r2.final_lambda
            assert r1.accepted == r2.accepted and r1.lm_attempts == r2.lm_attempts
            check("native RCCL collectives, world 1, distributed solve", "final cost rel", abs(r1.final_cost - r2.final_cost) / r1.final_cost, 5e-8)
        s1, s2 = e1.get_state(st), e2.get_state(st)
        check("native RCCL collectives, world 1, distributed solve", "points abs", np.abs(s1.points - s2.points).max(), 5e-8)
        e1.close(); e2.close()
    finally:
        rc.close()


@pytest.mark.parametrize("world", [2, 8])
def test_distributed_factorisation_at_the_benchmarked_size(tmp_path, world):
    """The distributed schedule with its DEFAULT parameters at the size of BASELINE configs[1] (D = 12 525, n_pad = 12 672: first band
    of 2048 rows all-reduced, 10 624 rows reduce-scattered into 25 column groups; two ranks: three super-panels and 6400 rows gathered
    for the final launch, eight ranks: four and 4352 -- the rank-dependent default of ldlt_tail_rows), two and EIGHT processes on one GPU (eight: 7-8 imagesets per rank, the 25 column groups dealt 4 / 3 to the ranks
    -- the ownership the first real 8-GPU run will have) with the collectives through a cba_collective_fn (host-staged gloo),
    three LM iterations against the single-process engine."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, True, True), nprocs=world, join=True)
    pb, st, _ = _problem_full_grid()
    en = eng.Engine(pb, deterministic=True, last_projection=pb.obs_xy.astype(np.float64))
    en.set_state(st)
    lam = -1.0
    reps = []
    for _ in range(STEPS):
        r = en.step(lam)
        lam = r.final_lambda
        reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
    ref = en.get_state(st)
    en.close()
    reps = np.array(reps)
    case = f"{world} ranks on 1 GPU, distributed factorisation at cfg-2 size (60 imagesets, 84x60 grid) vs single process"
    rk = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    for k in range(world):
        check_equal(case, f"rank {k}: LM attempts / accept decisions / valid counts", int(np.count_nonzero(rk[k]["reps"][:, 3:] != reps[:, 3:])))
        check(case, f"rank {k}: costs rel", (np.abs(rk[k]["reps"][:, :2] - reps[:, :2]) / reps[:, :2]).max(), 5e-7)
        b, e = int(rk[k]["b"]), int(rk[k]["e"])
        check(case, f"rank {k}: own poses abs", np.abs(rk[k]["poses"] - ref.rig_tr_global[b:e]).max(), 5e-8)
        check(case, f"rank {k}: points abs", np.abs(rk[k]["points"] - ref.points).max(), 5e-8)
        check(case, f"rank {k}: grid abs", np.abs(rk[k]["grid0"] - ref.grids[0]).max(), 1e-7)
    for key in ("points", "grid0"):
        for k in range(1, world):
            check_equal(case, f"replicated state identical on ranks 0 and {k}: {key}", int(np.count_nonzero(rk[0][key] != rk[k][key])))



def _rccl_worker(rank, world, port, out_dir, distributed_solve):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    eng.prepare(rank)
    pb, st, _ = _problem(DIST_GRID)
    shards = dist_mod.shard_images(np.bincount(pb.obs_image, minlength=pb.n_images), world)
    b, e = shards[rank]
    sub, sst = pb.image_slice(b, e), st.image_slice(b, e)
    rc = dist_mod.NativeRccl(rank, world, os.path.join(out_dir, f"rccl_id_{port}"), rank)
    en = eng.Engine(sub, device=rank, allreduce_native=(rc.fn, rc.user), n_images_global=pb.n_images, deterministic=True,
                    last_projection=sub.obs_xy.astype(np.float64), distributed_solve=distributed_solve, rank=rank, world_size=world,
                    collective_native=(rc.collective_fn, rc.user) if distributed_solve else None, factor_tail_rows=DIST_TAIL_ROWS)
    en.set_state(sst)
    lam = -1.0
    reps = []
    for _ in range(STEPS):
        r = en.step(lam)
        lam = r.final_lambda
        reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
    out = en.get_state(sst)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b=b, e=e, reps=np.array(reps), poses=out.rig_tr_global, points=out.points,
             grid0=out.grids[0], comm=rc.comm_count())
    en.close()
    rc.close()


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (skips on the 1-GPU lease, runs on a multi-GPU node)")
@pytest.mark.parametrize("distributed_solve", [False, True])
def test_native_rccl_two_ranks_on_two_gpus(tmp_path, distributed_solve):
    """The C++ host's path with a real communicator: libcalib_ba_rccl.so (ncclAllReduce, and ncclReduceScatter / ncclAllGather
    for the distributed solve) between two processes on two devices, three LM iterations against the single-process engine."""
    world = 2
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path), distributed_solve), nprocs=world, join=True)
    pb, st, _ = _problem(DIST_GRID)
    en = eng.Engine(pb, deterministic=True, last_projection=pb.obs_xy.astype(np.float64), factor_tail_rows=DIST_TAIL_ROWS)
    en.set_state(st)
    lam = -1.0
    reps = []
    for _ in range(STEPS):
        r = en.step(lam)
        lam = r.final_lambda
        reps.append([r.initial_cost, r.final_cost, r.final_lambda, r.lm_attempts, float(r.accepted), r.n_residuals_valid])
    ref = en.get_state(st)
    en.close()
    reps = np.array(reps)
    case = f"native RCCL, 2 ranks on 2 GPUs, {'distributed' if distributed_solve else 'replicated'} solve vs single process"
    rk = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    for k in range(world):
        check_equal(case, f"rank {k}: communicator size", int(rk[k]["comm"]) - world)
        check_equal(case, f"rank {k}: LM attempts / accept decisions / valid counts", int(np.count_nonzero(rk[k]["reps"][:, 3:] != reps[:, 3:])))
        check(case, f"rank {k}: costs rel", (np.abs(rk[k]["reps"][:, :2] - reps[:, :2]) / reps[:, :2]).max(), 5e-7)
        check(case, f"rank {k}: points abs", np.abs(rk[k]["points"] - ref.points).max(), 5e-8)
        check(case, f"rank {k}: grid abs", np.abs(rk[k]["grid0"] - ref.grids[0]).max(), 1e-7)
    check_equal(case, "replicated state identical on both ranks", int(np.count_nonzero(rk[0]["points"] != rk[1]["points"])))
