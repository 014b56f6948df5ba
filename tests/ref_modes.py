"""Accumulation modes pinned against the reference's UpdateEquationAccumulator (tests/golden/ref_accumulated.npz).

name -> (committed first-iteration fixture the problem comes from, flag overrides).  Shared by the generator
(tests/golden/make_ref_fixtures.py) and the tests; rows B2 / B3 / A5 of SURVEY 8(a): every (localize_only x eliminate_points x
rig) branch of AccumulateModelJacobian (APP/bundle_adjustment/joint_optimization.cc:479-590) and both intrinsics block
sizes (32 central, 80 non-central)."""
ACC_MODES = {
    "central": ("first_iteration_1cam_central", {}),
    "rig": ("first_iteration_2cam_central", {}),
    "noncentral": ("first_iteration_1cam_noncentral", {}),
    "eliminate": ("first_iteration_eliminate_points", {}),
    "eliminate_rig": ("first_iteration_2cam_central", dict(eliminate_points=True)),
    "localize": ("first_iteration_1cam_central", dict(localize_only=True)),
    "localize_rig": ("first_iteration_2cam_central", dict(localize_only=True)),
    "localize_eliminate": ("first_iteration_1cam_central", dict(localize_only=True, eliminate_points=True)),
    "localize_eliminate_rig": ("first_iteration_2cam_central", dict(localize_only=True, eliminate_points=True)),
}
ACC_FIELDS = ("block_diag_H", "off_diag_H", "dense_H", "block_diag_b", "dense_b")


def load_mode(name, golden_dir):
    import os
    from tests.test_golden_fixtures import load_fixture
    fixture, over = ACC_MODES[name]
    pb, st, _ = load_fixture(os.path.join(golden_dir, fixture + ".npz"))
    for k, v in over.items():
        setattr(pb, k, v)
    return pb, st
