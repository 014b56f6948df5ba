"""Generates golden vectors from the REFERENCE ITSELF (oracle/_ref = reference sources compiled by oracle/Makefile).

Run in the build container (needs /root/reference):   python tests/golden/make_ref_fixtures.py
Outputs (committed):
  tests/golden/ref_generic_models_17x13.yaml   the real calibrated 17 x 13 central-generic camera that the reference
                                               holds as a test vector (generic_models/src/main.cc:86-98), verbatim
  tests/golden/ref_vectors.npz                 inputs + outputs of the reference's own functions on seeded inputs:
                                               Unproject / UnprojectWithJacobian / Project of both generic models,
                                               ComputeJacobian / ComputeRigJacobian, the generated un-projection patches,
                                               tangents, local updates, quaternion update, Huber loss, B-spline surface
The .npz lets the CPU suite pin the oracle, and the GPU suite pin the HIP kernels, against reference-computed numbers
on machines where /root/reference does not exist.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from camera_calibration_amd.problem import CENTRAL_GENERIC, NONCENTRAL_GENERIC, Camera  # noqa: E402
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MAIN_CC = "/root/reference/applications/camera_calibration/generic_models/src/main.cc"


def extract_yaml() -> str:
    src = open(MAIN_CC, encoding="utf-8").read()
    blocks = re.findall(r'R"yaml\((.*?)\)yaml"', src, flags=re.S)
    central = [b for b in blocks if b.startswith("type : CentralGenericModel")]
    assert len(central) == 1
    return central[0]


def write_reference_dataset_bin(L, path):
    """A dataset.bin whose every field is written by the reference's own write_one overloads (APP/io/io_util.h:37-69),
    called in the order of SaveDataset (APP/io/calibration_io.cc:51-137).  SaveDataset itself cannot be compiled here (Qt,
    boost, yaml-cpp); the content is the example dataset of tests/test_calibration_io.py."""
    import ctypes as C
    L.ref_io_append.argtypes = [C.c_char_p, C.c_int, C.c_double]
    U32, I32, F32 = 0, 1, 2
    bp = path.encode()
    def raw(b):
        with open(path, "ab") as f:
            f.write(b)
    def put(kind, v):
        assert L.ref_io_append(bp, kind, float(v)) == 0
    if os.path.exists(path):
        os.remove(path)
    raw(b"calib_data"); put(U32, 0)
    put(U32, 2)
    for w, h in ((640, 480), (800, 600)):
        put(U32, w); put(U32, h)
    f0 = [(1.5, 2.25, 7), (3.0, 4.0, -2)]; f1 = [(10.125, 20.5, 123456)]
    put(U32, 2)
    for name, per_camera in ((b"img_000.png", (f0, f1)), (b"", ((), f1))):
        put(U32, len(name)); raw(name)
        for feats in per_camera:
            put(U32, len(feats))
            for x, y, fid in feats:
                put(F32, x); put(F32, y); put(I32, fid)
    put(U32, 1)
    put(F32, 0.012); put(U32, 2)
    for fid, (x, y) in ((7, (1, 2)), (-2, (0, -3))):
        put(I32, fid); put(I32, x); put(I32, y)


def write_accumulated(path):
    """B2 / B3 / A5: the oracle's per-observation records fed through the REFERENCE's accumulator (ref.accumulate_records) for every
    mode of tests/ref_modes.py, and -- M5 on a whole problem -- the reference's ProjectionJacobianWrtIntrinsics at every observation
    of the central single-camera fixture."""
    from oracle import oracle as orc
    from tests.ref_modes import ACC_FIELDS, ACC_MODES, load_mode
    out = {}
    for name in ACC_MODES:
        pb, st = load_mode(name, HERE)
        op = orc.OracleProblem(pb)
        _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
        r = ref.accumulate_records(pb, recs)
        for f in ACC_FIELDS:
            out[f"{name}__{f}"] = r[f]
        out[f"{name}__cost"] = np.float64(r["cost"])
        out[f"{name}__cost_vector"] = r["cost_vector"]
    pb, st = load_mode("central", HERE)
    op = orc.OracleProblem(pb)
    _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
    cam = pb.cameras[0]
    idx = np.full((pb.n_obs, 32), -1, dtype=np.int32); jac = np.zeros((pb.n_obs, 2, 32)); ok = np.zeros(pb.n_obs, dtype=np.int32)
    for o in range(pb.n_obs):
        if not recs[o].valid:
            continue
        itg = orc.se3_mul(st.camera_tr_rig[0], st.rig_tr_global[int(pb.obs_image[o])])
        q = itg[:4]; w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        local = R @ st.points[int(pb.obs_point[o])] + itg[4:]
        ok[o], idx[o], jac[o] = ref.central_grid_projection_jacobian(cam, st.grids[0], local, np.array(recs[o].pixel[:]), pb.fd_delta)
    out.update(central__m5_ok=ok, central__m5_idx=idx, central__m5_jac=jac)
    # N3: the same for the non-central fixture (80 parameters per observation)
    pb, st = load_mode("noncentral", HERE)
    op = orc.OracleProblem(pb)
    _, _, recs = op.jacobian_pass(st, op.new_system(), want_records=True)
    cam = pb.cameras[0]
    idx = np.full((pb.n_obs, 80), -1, dtype=np.int32); jac = np.zeros((pb.n_obs, 2, 80)); ok = np.zeros(pb.n_obs, dtype=np.int32)
    for o in range(pb.n_obs):
        if not recs[o].valid:
            continue
        itg = orc.se3_mul(st.camera_tr_rig[0], st.rig_tr_global[int(pb.obs_image[o])])
        w, x, y, z = itg[:4]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        local = R @ st.points[int(pb.obs_point[o])] + itg[4:]
        ok[o], idx[o], jac[o] = ref.noncentral_grid_projection_jacobian(cam, st.grids[0], local, np.array(recs[o].pixel[:]), pb.fd_delta)
    out.update(noncentral__m5_ok=ok, noncentral__m5_idx=idx, noncentral__m5_jac=jac)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


def main():
    L = ref.lib()
    dp = ref._dp
    rng = np.random.default_rng(20260924)
    out = {}
    yaml_path = os.path.join(HERE, "ref_generic_models_17x13.yaml")
    with open(yaml_path, "w", encoding="utf-8") as f:
        f.write(extract_yaml())
    cam17, params, grid17 = ref.RefCamera.read_yaml(yaml_path)
    out["c17_params"] = np.array(params, dtype=np.int32)          # width height min_x min_y max_x max_y gw gh
    out["c17_grid"] = grid17.reshape(-1, 3)
    # --- central 17 x 13: the reference's own round trip (main.cc:38-84 pattern) on the calibrated area ---
    n = 400
    px = np.stack([rng.uniform(params[2], params[4] + 1, n), rng.uniform(params[3], params[5] + 1, n)], axis=1)
    dirs = np.zeros((n, 3)); jac = np.zeros((n, 3, 2)); ok = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        o, d, j = cam17.unproject(px[i], jacobian=True)
        ok[i] = o; dirs[i] = d; jac[i] = j
    assert ok.all()
    pts = dirs * rng.uniform(0.3, 4.0, (n, 1))
    reproj = np.zeros((n, 2)); pok = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        o, p = cam17.project(pts[i])
        pok[i] = o; reproj[i] = p
    # warm-started projection from a perturbed estimate (ProjectWithInitialEstimate)
    init = np.clip(px + rng.normal(0, 6.0, px.shape), [params[2], params[3]], [params[4] + 0.9, params[5] + 0.9])
    reproj_init = np.zeros((n, 2)); pok_init = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        o, p = cam17.project(pts[i], init=init[i])
        pok_init[i] = o; reproj_init[i] = p
    # points that do not project (behind / far outside): the reference returns false
    bad = np.array([[0.0, 0.0, -1.0], [50.0, 0.0, 1.0], [0.0, -80.0, 1.0], [-3.0, -3.0, 0.1]])
    bad_ok = np.array([cam17.project(b)[0] for b in bad], dtype=np.uint8)
    out.update(c17_px=px, c17_dirs=dirs, c17_jac=jac, c17_pts=pts, c17_reproj=reproj, c17_reproj_ok=pok, c17_init=init,
               c17_reproj_init=reproj_init, c17_reproj_init_ok=pok_init, c17_bad_pts=bad, c17_bad_ok=bad_ok)
    # --- non-central 8 x 8, the grid of TestNoncentralGenericCameraReprojection (main.cc:146-160) ---
    gw = gh = 8
    gy, gx = np.meshgrid(np.arange(gh, dtype=np.float64), np.arange(gw, dtype=np.float64), indexing="ij")
    d = np.stack([gx, gy, np.ones_like(gx)], axis=-1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.stack([0.2 * gx, 0.1 * gy, 0.01 * gx], axis=-1).reshape(-1, 3)
    ncam = Camera(NONCENTRAL_GENERIC, 640, 480, 0, 0, 639, 479, gw, gh)
    ngrid = np.stack([d, o])
    nref = ref.RefCamera(ncam, ngrid)
    m = 300
    npx = np.stack([rng.uniform(0, 640, m), rng.uniform(0, 480, m)], axis=1)
    nlines = np.zeros((m, 6)); njac = np.zeros((m, 6, 2))
    for i in range(m):
        okk, l, j = nref.unproject(npx[i], jacobian=True)
        assert okk
        nlines[i] = l; njac[i] = j
    npts = nlines[:, 3:] + nlines[:, :3] * rng.uniform(0.5, 30.0, (m, 1))
    nreproj = np.zeros((m, 2)); nok = np.zeros(m, dtype=np.uint8)
    for i in range(m):
        okk, p = nref.project(npts[i])
        nok[i] = okk; nreproj[i] = p
    out.update(n8_grid=ngrid, n8_px=npx, n8_lines=nlines, n8_jac=njac, n8_pts=npts, n8_reproj=nreproj, n8_reproj_ok=nok)
    # --- generated Jacobians of the problem layer (joint_optimization_jacobians.h) ---
    k = 64
    q = rng.normal(size=(k, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q2 = rng.normal(size=(k, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    p3 = rng.normal(size=(k, 3)) * 2.0
    t3 = rng.normal(size=(k, 3))
    j30 = np.zeros((k, 30)); j51 = np.zeros((k, 51))
    for i in range(k):
        L.ref_compute_jacobian(dp(q[i].copy()), dp(p3[i].copy()), dp(j30[i]))
        L.ref_compute_rig_jacobian(dp(q[i].copy()), dp(p3[i].copy()), dp(q2[i].copy()), dp(t3[i].copy()), dp(j51[i]))
    out.update(jac_q=q, jac_q2=q2, jac_p=p3, jac_t=t3, jac_30=j30, jac_51=j51)
    # --- generated un-projection patches (central_generic_jacobians.cc / noncentral_generic_jacobians.cc) ---
    frac = rng.uniform(3.0, 4.0, (k, 2))
    cp = rng.normal(size=(k, 16, 3)); cp[:, :, 2] += 4.0
    cdir = np.zeros((k, 3)); cj = np.zeros((k, 6))
    lp = np.concatenate([cp, rng.normal(size=(k, 16, 3)) * 0.01], axis=2)
    lline = np.zeros((k, 6)); lj = np.zeros((k, 12))
    for i in range(k):
        L.ref_central_unproject_patch(frac[i, 0], frac[i, 1], dp(np.ascontiguousarray(cp[i]).ravel()), dp(cdir[i]), dp(cj[i]))
        L.ref_noncentral_unproject_patch(frac[i, 0], frac[i, 1], dp(np.ascontiguousarray(lp[i]).ravel()), dp(lline[i]), dp(lj[i]))
    out.update(patch_frac=frac, patch_central=cp, patch_central_dir=cdir, patch_central_jac=cj, patch_lines=lp,
               patch_line_out=lline, patch_line_jac=lj)
    # --- parametrisations ---
    dd = rng.normal(size=(k, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    dd[0] = [1.0, 0.0, 0.0]; dd[1] = [0.95, 0.2, np.sqrt(1 - 0.95 ** 2 - 0.04)]; dd[2] = [0.0, 0.0, 1.0]
    t1 = np.zeros((k, 3)); t2 = np.zeros((k, 3)); tj = np.zeros((k, 18)); lu = np.zeros((k, 6))
    off = rng.normal(size=(k, 5)) * 0.05
    dirupd = np.zeros((k, 3)); lines_in = np.concatenate([dd, rng.normal(size=(k, 3))], axis=1); lineupd = np.zeros((k, 6))
    for i in range(k):
        L.ref_tangents(dp(dd[i].copy()), dp(t1[i]), dp(t2[i]))
        L.ref_tangents_jacobian(dp(dd[i].copy()), dp(tj[i]))
        L.ref_local_update_jacobian_wrt_direction(dp(dd[i].copy()), dp(lu[i]))
        L.ref_apply_direction_update(dp(dd[i].copy()), off[i, 0], off[i, 1], dp(dirupd[i]))
        L.ref_apply_line_update(dp(lines_in[i].copy()), dp(off[i].copy()), dp(lineupd[i]))
    out.update(par_dir=dd, par_t1=t1, par_t2=t2, par_tangent_jac=tj, par_local_jac=lu, par_offsets=off, par_dir_updated=dirupd,
               par_lines=lines_in, par_lines_updated=lineupd)
    upd = rng.normal(size=(k, 3)) * np.logspace(-6, 0, k)[:, None]
    upd[0] = 0.0
    qout = np.zeros((k, 4)); qj = np.zeros((k, 12))
    for i in range(k):
        L.ref_apply_quaternion_update(dp(q[i].copy()), dp(upd[i].copy()), dp(qout[i]))
        L.ref_quaternion_jacobian(dp(q[i].copy()), dp(qj[i]))
    out.update(quat_update=upd, quat_out=qout, quat_jac=qj)
    # --- Huber ---
    sq = np.concatenate([np.array([0.0, 0.25, 0.999999, 1.0, 1.000001, 4.0]), rng.uniform(0, 9, 20)])
    out["huber_sq"] = sq
    out["huber_cost_sq"] = np.array([L.ref_huber_cost_sq(v, 1.0) for v in sq])
    out["huber_weight_sq"] = np.array([L.ref_huber_weight_sq(v, 1.0) for v in sq])
    # --- B-spline surface: the fixed control net of BSpline.SlowFastAlgorithmConsistency (APP/test/b_spline_test.cc:41-47) ---
    net = np.array([[0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [1, 1], [2, 2], [3, 3], [0, 0], [4, 4], [5, 5], [6, 6], [0, 0], [7, 7],
                    [8, 8], [9, 9]], dtype=np.float32)
    xs = 1.0 + np.arange(500) / 500.0
    fast = np.zeros((500, 2), dtype=np.float32); slow = np.zeros((500, 2), dtype=np.float32)
    fast64 = np.zeros((500, 2)); slow64 = np.zeros((500, 2))
    net64 = net.astype(np.float64)
    import ctypes as C
    fpt = C.POINTER(C.c_float)
    for i, x in enumerate(xs):
        L.ref_bspline_surface_f32(net.ctypes.data_as(fpt), 4, 4, x, 1.5, 0, fast[i].ctypes.data_as(fpt))
        L.ref_bspline_surface_f32(net.ctypes.data_as(fpt), 4, 4, x, 1.5, 1, slow[i].ctypes.data_as(fpt))
        L.ref_bspline_surface(dp(net64.ravel()), 4, 4, 2, x, 1.5, 0, dp(fast64[i]))
        L.ref_bspline_surface(dp(net64.ravel()), 4, 4, 2, x, 1.5, 1, dp(slow64[i]))
    assert np.abs(fast - slow).max() <= 1e-5          # the reference's own assertion
    out.update(bsp_net=net64, bsp_x=xs, bsp_fast_f32=fast, bsp_slow_f32=slow, bsp_fast=fast64, bsp_slow=slow64)
    # --- round 3: CentralGridModel::ProjectionJacobianWrtIntrinsics / SubtractDelta (APP/models/central_grid.h:168-245) on the 17 x 13 camera ---
    c17 = Camera(CENTRAL_GENERIC, *params)
    m5n = 80
    m5_ok = np.zeros(m5n, dtype=np.int32); m5_idx = np.zeros((m5n, 32), dtype=np.int32); m5_jac = np.zeros((m5n, 2, 32))
    for i in range(m5n):
        m5_ok[i], m5_idx[i], m5_jac[i] = ref.central_grid_projection_jacobian(c17, grid17, pts[i], reproj[i], 1e-4)
    m6_delta = rng.normal(size=2 * params[6] * params[7]) * 0.01
    out.update(m5_pts=pts[:m5n], m5_px=reproj[:m5n], m5_delta=np.float64(1e-4), m5_ok=m5_ok, m5_idx=m5_idx, m5_jac=m5_jac,
               m6_delta=m6_delta, m6_grid=ref.central_grid_subtract_delta(c17, grid17, m6_delta))
    # N3 SubtractDelta on the non-central 8 x 8 self-test camera
    n3_delta = rng.normal(size=5 * gw * gh) * 0.01
    out.update(n3_delta=n3_delta, n3_grids=ref.noncentral_grid_subtract_delta(ncam, ngrid, n3_delta))
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
    write_accumulated(os.path.join(HERE, "ref_accumulated.npz"))
    write_reference_dataset_bin(L, os.path.join(HERE, "ref_dataset.bin"))
    print("wrote", yaml_path, "and ref_vectors.npz:", {k_: v.shape for k_, v in out.items()})


if __name__ == "__main__":
    main()
