import sys, time, traceback
sys.path.insert(0, '.')
import numpy as np
from camera_calibration_amd import engine as eng, synthetic as syn
from camera_calibration_amd.problem import *
from oracle import oracle as orc
np.set_printoptions(linewidth=200, precision=6)
print(eng.load().cba_version())
def oproj(cam, grid, pts): return orc.project(cam, grid, pts)
try:
    nan=float('nan')
    x = eng.schur_solve(np.array([[[1,5],[nan,6]],[[9,5],[nan,4]]],float), np.array([[3,4],[7,8],[7,6],[3,2]],float), np.array([[1,4],[nan,7]],float), np.array([1.,2,3,4]), np.array([5.,6]))
    print('schur golden', x)
except Exception: traceback.print_exc()
try:
    pb, st, gt = syn.reference_test_problem(1, oproj, seed=0)
    op = orc.OracleProblem(pb)
    sysm = op.new_system()
    c_ref, v_ref, recs = op.jacobian_pass(st, sysm, want_records=True)
    e = eng.Engine(pb); e.set_state(st)
    t=time.time(); c = e.debug_accumulate(); print('accumulate', time.time()-t, c, c_ref)
    v = e.dump(eng.DUMP_COST_VECTOR); print('costvec maxdiff', np.abs(v-v_ref).max(), 'valid mismatch', (v>=0).__ne__(v_ref>=0).sum())
    fl = e.dump(eng.DUMP_FLAGS); print('flags hist', np.bincount(fl, minlength=4))
    pix = e.dump(eng.DUMP_PIXELS); pr = np.array([[r.pixel[0], r.pixel[1]] for r in recs]); print('pix maxdiff', np.abs(pix-pr).max())
    J = e.dump(eng.DUMP_JACOBIANS)
    r0 = recs[0]
    print('J[0] pose gpu', J[0,3:15]); print('J[0] pose ref', np.array(r0.pose_jac[:]))
    print('J[0] pt gpu', J[0,27:33]); print('J[0] pt ref', np.array(r0.point_jac[:]))
    print('J[0] grid gpu', J[0,33:41]); print('J[0] grid ref', np.array(r0.grid_jac[:8]))
    for name, a, b in [('bD', e.dump(eng.DUMP_BLOCK_DIAG_H), sysm.block_diag_H), ('bb', e.dump(eng.DUMP_BLOCK_DIAG_B), sysm.block_diag_b), ('off', e.dump(eng.DUMP_OFF_DIAG_H), sysm.off_diag_H), ('dH', np.triu(e.dump(eng.DUMP_DENSE_H)), np.triu(sysm.dense_H)), ('db', e.dump(eng.DUMP_DENSE_B), sysm.dense_b)]:
        if name=='bD': a=np.array([np.triu(x) for x in a]); b=np.array([np.triu(x) for x in b])
        print(name, 'maxabs', np.abs(b).max(), 'maxdiff', np.abs(a-b).max())
    lam = 1e-5*(np.trace(sysm.dense_H)+sum(np.trace(b) for b in sysm.block_diag_H))/pb.total_dof
    sysm.add_lambda(lam); xr = orc.schur_solve(sysm)
    xg = e.debug_solve(lam); print('x maxabs', np.abs(xr).max(), 'maxdiff', np.abs(xg-xr).max())
    xs = eng.schur_solve(sysm.block_diag_H, sysm.off_diag_H, sysm.dense_H, sysm.block_diag_b, sysm.dense_b); print('solver-only diff', np.abs(xs-xr).max())
    e.close()
    e = eng.Engine(pb); e.set_state(st); st_ref = st.copy(); lam=lr=-1
    for i in range(12):
        t=time.time(); rep = e.step(lam); dt=time.time()-t; r = op.optimize_jointly(st_ref, 1, lr); lam=rep.final_lambda; lr=r['final_lambda']
        print(i, 'gpu', rep.final_cost, rep.final_lambda, rep.lm_attempts, rep.accepted, 'orc', r['cost'], r['final_lambda'], r['lm_attempts'], 'dt %.3f tj %.3f ts %.3f tc %.3f'%(dt, rep.t_jac, rep.t_solve, rep.t_cost))
    e.close()
except Exception: traceback.print_exc()
