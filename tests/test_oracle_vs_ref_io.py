"""SURVEY 8f row F2 pinned to the REFERENCE'S OWN CODE: SaveDataset / LoadDataset, SaveCameraModel, SavePoses and
SavePointsAndIndexMapping (APP/io/calibration_io.cc:51-246, 526-647, 785-839, 890-937), piped from /root/reference into
oracle/_ref/libcalibref_ba.so (oracle/Makefile, oracle/ref_ba_glue.cc); the YAML readers of that file need yaml-cpp and are not
compiled.  Rounds 1-4 pinned dataset.bin to a file written through the reference's write_one primitives and the YAML side to a
round trip through PyYAML.

* dataset.bin: what the product writes is loaded by the reference's LoadDataset and written back by its SaveDataset: identical bytes
  (the product's file is a fixed point of the reference's reader + writer), with the counts the reference saw;
* the YAML writers: byte-identical files for the same data (std::setprecision(14) = %.14g), incl. the .obj files the reference always
  writes next to poses and points; the feature_id_to_point_index list is compared as a set -- the reference walks an unordered_map."""
import os

import numpy as np
import pytest

from camera_calibration_amd import calibration_io as cio
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import NONCENTRAL_GENERIC
from oracle import oracle as orc
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.ba_available(), reason="oracle/_ref/libcalibref_ba.so not built (needs /root/reference)")


def _dataset(rng, n_cameras=2, n_imagesets=5):
    ds = cio.DatasetData(image_sizes=[(640 + 16 * c, 480 + 8 * c) for c in range(n_cameras)])
    for i in range(n_imagesets):
        feats = []
        for c in range(n_cameras):
            n = int(rng.integers(0, 40)) if (i + c) % 4 else 0          # some cameras see nothing in some imagesets
            f = np.zeros(n, dtype=cio.FEATURE_DTYPE)
            f["x"] = rng.uniform(0, 640, n).astype(np.float32); f["y"] = rng.uniform(0, 480, n).astype(np.float32)
            f["id"] = rng.integers(-5, 2000, n)
            feats.append(f)
        ds.imagesets.append(cio.ImagesetData("" if i == 2 else f"image_{i:04d}.png", feats))
    ds.known_geometries.append(cio.KnownGeometry(0.0125, {int(k): (int(k) % 17 - 8, int(k) // 17) for k in rng.integers(0, 500, 60)}))
    ds.known_geometries.append(cio.KnownGeometry(0.03, {}))
    return ds


def test_dataset_bin_is_a_fixed_point_of_the_references_reader_and_writer(tmp_path):
    rng = np.random.default_rng(0)
    for case in range(3):
        ds = _dataset(rng, n_cameras=1 + case, n_imagesets=3 + 4 * case)
        a, b = str(tmp_path / f"a{case}.bin"), str(tmp_path / f"b{case}" / "dataset.bin")
        cio.save_dataset(a, ds)
        seen = ref.f2_dataset_load_and_save(a, b)
        assert seen is not None
        assert seen["cameras"] == len(ds.image_sizes) and seen["imagesets"] == len(ds.imagesets) and seen["known_geometries"] == 2
        assert seen["features"] == sum(len(f) for s in ds.imagesets for f in s.features)
        ba, bb = open(a, "rb").read(), open(b, "rb").read()
        # identical up to the ORDER of the (feature id -> position) entries of a known geometry: the reference walks an unordered_map
        # (calibration_io.cc:116-127), so the entries come out in libstdc++'s bucket order; everything in front of them is byte-identical
        entries = sum(12 * len(g.feature_id_to_position) for g in ds.known_geometries)
        tail = entries + 8 * len(ds.known_geometries)                     # + cell length and entry count per geometry
        assert len(ba) == len(bb) and ba[:len(ba) - tail] == bb[:len(bb) - tail]
        back = cio.load_dataset(b)                                         # and the product reads what the reference wrote
        for g0, g1 in zip(ds.known_geometries, back.known_geometries):
            assert g1.feature_id_to_position == g0.feature_id_to_position and g1.cell_length_in_meters == np.float32(g0.cell_length_in_meters)
        assert back.image_sizes == ds.image_sizes and [s.filename for s in back.imagesets] == [s.filename for s in ds.imagesets]
        for s0, s1 in zip(ds.imagesets, back.imagesets):
            for f0, f1 in zip(s0.features, s1.features):
                np.testing.assert_array_equal(f0, f1)


def test_dataset_bin_with_single_entry_geometries_is_byte_identical(tmp_path):
    rng = np.random.default_rng(1)
    ds = _dataset(rng, n_cameras=2, n_imagesets=6)
    ds.known_geometries = [cio.KnownGeometry(0.02, {41: (3, -2)}), cio.KnownGeometry(0.5, {7: (0, 0)})]
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    cio.save_dataset(a, ds)
    assert ref.f2_dataset_load_and_save(a, b) is not None
    assert open(a, "rb").read() == open(b, "rb").read()


def test_dataset_bin_bad_files_are_rejected_by_both(tmp_path):
    p = str(tmp_path / "bad.bin")
    open(p, "wb").write(b"calib_dat" + b"\x00" * 40)
    assert ref.f2_dataset_load_and_save(p, str(tmp_path / "out.bin")) is None
    with pytest.raises(Exception):
        cio.load_dataset(p)


@pytest.mark.parametrize("model_type", [0, NONCENTRAL_GENERIC])
def test_camera_model_yaml_is_the_references(tmp_path, model_type):
    pb, st, _ = syn.reference_test_problem(1, orc.project, seed=3, num_points=10, num_poses=2, model_type=model_type)
    cam, grid = pb.cameras[0], st.grids[0]
    a, b = str(tmp_path / "a.yaml"), str(tmp_path / "sub" / "b.yaml")
    cio.save_camera_model(a, cam, grid)
    assert ref.f2_save_camera_model(cam, grid, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    cam2, grid2 = cio.load_camera_model(b)                                 # the product reads the reference's file
    assert cam2 == cam
    np.testing.assert_allclose(np.asarray(grid2).reshape(-1), np.asarray(grid).reshape(-1), rtol=0, atol=1e-13)


def test_poses_yaml_and_obj_are_the_references(tmp_path):
    pb, st, _ = syn.reference_test_problem(2, orc.project, seed=5, num_points=10, num_poses=7)
    used = np.array([1, 1, 0, 1, 0, 1, 1], dtype=bool)
    a, b = str(tmp_path / "a.yaml"), str(tmp_path / "b.yaml")
    cio.save_poses(a, used, st.rig_tr_global)
    assert ref.f2_save_poses(used, st.rig_tr_global, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # the .obj of camera centres -R^T t: same lines, the 14th digit may differ (another order of the same products)
    oa, ob = np.loadtxt(a + ".obj", usecols=(1, 2, 3, 4, 5, 6)), np.loadtxt(b + ".obj", usecols=(1, 2, 3, 4, 5, 6))
    assert oa.shape == ob.shape == (int(used.sum()), 6) and (oa[:, 3:] == [1, 0, 0]).all()
    np.testing.assert_allclose(oa, ob, rtol=0, atol=1e-12)
    used2, poses2 = cio.load_poses(b)
    np.testing.assert_array_equal(used2, used)
    np.testing.assert_allclose(poses2[used], st.rig_tr_global[used], rtol=0, atol=1e-13)


def test_points_yaml_and_obj_are_the_references(tmp_path):
    rng = np.random.default_rng(2)
    pts = rng.normal(0, 0.3, (40, 3))
    mapping = {int(f): i for i, f in enumerate(rng.permutation(5000)[:40])}
    a, b = str(tmp_path / "a.yaml"), str(tmp_path / "b.yaml")
    cio.save_points(a, pts, mapping)
    assert ref.f2_save_points(pts, mapping, b)
    ta, tb = open(a).read(), open(b).read()
    head_a, list_a = ta.split("feature_id_to_point_index:\n"); head_b, list_b = tb.split("feature_id_to_point_index:\n")
    assert head_a == head_b                                               # comment + the points line, byte for byte
    import re
    entries = lambda t: sorted(re.findall(r"  - feature_id: (-?\d+)\n    point_index: (-?\d+)\n", t))     # noqa: E731  (the reference iterates an unordered_map)
    assert entries(list_a) == entries(list_b) and len(entries(list_a)) == 40
    assert len(list_a) == len(list_b)                                     # nothing but those entries, in the same line format
    assert open(a + ".obj", "rb").read() == open(b + ".obj", "rb").read()
    pts2, mapping2 = cio.load_points(b)
    assert mapping2 == mapping
    np.testing.assert_allclose(pts2, pts, rtol=1e-13)
