"""on-disk formats (APP/io/calibration_io.cc:51-247, 432-985).

* dataset.bin: byte-for-byte against a file assembled by hand from the format description (magic, big-endian
  u32/i32 through htonl, raw little-endian f32) -- the golden vector for this row;
* BAState YAML directory: independent parse with PyYAML, 14-significant-digit round trip, re-normalisation of
  direction grids / quaternions on load, line format;
* the C++ mirror (camera_calibration_amd/host/calibration_io.h) loads what Python wrote and writes it back
  (no GPU needed: the I/O functions make no HIP calls).
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest
import yaml

from camera_calibration_amd import calibration_io as cio
from camera_calibration_amd import synthetic as syn
from camera_calibration_amd.problem import NONCENTRAL_GENERIC
from oracle import oracle as orc


def _example_dataset():
    ds = cio.DatasetData(image_sizes=[(640, 480), (800, 600)])
    f0 = np.array([(1.5, 2.25, 7), (3.0, 4.0, -2)], dtype=cio.FEATURE_DTYPE)
    f1 = np.array([(10.125, 20.5, 123456)], dtype=cio.FEATURE_DTYPE)
    ds.imagesets.append(cio.ImagesetData("img_000.png", [f0, f1]))
    ds.imagesets.append(cio.ImagesetData("", [np.zeros(0, dtype=cio.FEATURE_DTYPE), f1]))
    ds.known_geometries.append(cio.KnownGeometry(0.012, {7: (1, 2), -2: (0, -3)}))
    return ds


def _golden_bytes():
    b = b"calib_data" + struct.pack(">I", 0)
    b += struct.pack(">I", 2) + struct.pack(">II", 640, 480) + struct.pack(">II", 800, 600)
    b += struct.pack(">I", 2)
    b += struct.pack(">I", 11) + b"img_000.png"
    b += struct.pack(">I", 2) + struct.pack("<ff", 1.5, 2.25) + struct.pack(">i", 7) + struct.pack("<ff", 3.0, 4.0) + struct.pack(">i", -2)
    b += struct.pack(">I", 1) + struct.pack("<ff", 10.125, 20.5) + struct.pack(">i", 123456)
    b += struct.pack(">I", 0)
    b += struct.pack(">I", 0)
    b += struct.pack(">I", 1) + struct.pack("<ff", 10.125, 20.5) + struct.pack(">i", 123456)
    b += struct.pack(">I", 1) + struct.pack("<f", 0.012) + struct.pack(">I", 2)
    b += struct.pack(">iii", 7, 1, 2) + struct.pack(">iii", -2, 0, -3)
    return b


def test_dataset_bin_golden_bytes(tmp_path):
    p = str(tmp_path / "sub" / "dataset.bin")
    cio.save_dataset(p, _example_dataset())
    assert open(p, "rb").read() == _golden_bytes()
    ds = cio.load_dataset(p)
    assert ds.image_sizes == [(640, 480), (800, 600)]
    assert [s.filename for s in ds.imagesets] == ["img_000.png", ""]
    assert ds.imagesets[0].features[0]["id"].tolist() == [7, -2]
    assert ds.imagesets[0].features[0]["x"].tolist() == [1.5, 3.0]
    assert ds.imagesets[1].features[0].shape == (0,)
    assert ds.known_geometries[0].feature_id_to_position == {7: (1, 2), -2: (0, -3)}
    assert ds.known_geometries[0].cell_length_in_meters == np.float32(0.012)


def test_dataset_bin_matches_the_file_written_by_the_reference_primitives(tmp_path):
    """tests/golden/ref_dataset.bin was written field by field by the reference's own write_one overloads (APP/io/io_util.h,
    compiled into oracle/_ref) in SaveDataset's order (tests/golden/make_ref_fixtures.py): same content, same bytes."""
    want = open(os.path.join(os.path.dirname(__file__), "golden", "ref_dataset.bin"), "rb").read()
    assert want == _golden_bytes()
    p = str(tmp_path / "dataset.bin")
    cio.save_dataset(p, _example_dataset())
    assert open(p, "rb").read() == want
    ds = cio.load_dataset(os.path.join(os.path.dirname(__file__), "golden", "ref_dataset.bin"))
    assert ds.imagesets[0].features[1]["id"].tolist() == [123456] and ds.imagesets[0].features[0]["y"].tolist() == [2.25, 4.0]


def test_dataset_bin_rejects_bad_input(tmp_path):
    p = str(tmp_path / "bad.bin")
    open(p, "wb").write(b"calib_datX" + b"\0" * 16)
    with pytest.raises(ValueError):
        cio.load_dataset(p)
    open(p, "wb").write(b"calib_data" + struct.pack(">I", 1))
    with pytest.raises(ValueError):
        cio.load_dataset(p)


def _state(model_type=0, ncam=1):
    kw = dict(seed=5, num_points=12, num_poses=5)
    if model_type:
        kw["model_type"] = model_type
    pb, st, _ = syn.reference_test_problem(ncam, orc.project, **kw)
    return pb, st


@pytest.mark.parametrize("model_type", [0, NONCENTRAL_GENERIC])
def test_ba_state_yaml_round_trip(tmp_path, model_type):
    pb, st = _state(model_type)
    used = np.array([True, True, False, True, True])
    mapping = {100 + i: i for i in range(pb.n_points)}
    base = str(tmp_path / "state")
    cio.save_ba_state(base, used, pb.cameras, st, mapping)
    # independent parse
    rig = yaml.safe_load(open(os.path.join(base, "rig_tr_global.yaml")))
    assert rig["pose_count"] == 5 and [p["index"] for p in rig["poses"]] == [0, 1, 3, 4]
    assert abs(rig["poses"][2]["tx"] - st.rig_tr_global[3, 4]) <= 1e-13 * max(1.0, abs(st.rig_tr_global[3, 4]))
    intr = yaml.safe_load(open(os.path.join(base, "intrinsics0.yaml")))
    assert intr["type"] == ("NoncentralGenericModel" if model_type else "CentralGenericModel")
    assert intr["grid_width"] == pb.cameras[0].grid_w and intr["calibration_max_x"] == pb.cameras[0].calib_max_x
    first = open(os.path.join(base, "intrinsics0.yaml")).read().split("\n")[0]
    assert first == "type : " + intr["type"]
    assert open(os.path.join(base, "points.yaml")).read().startswith("# Each point is stored as x, y, z.\npoints : [")
    # library round trip
    used2, cams2, st2, map2 = cio.load_ba_state(base)
    np.testing.assert_array_equal(used2, used)
    assert cams2 == pb.cameras and map2 == mapping
    np.testing.assert_allclose(st2.points, st.points, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(st2.rig_tr_global[used], st.rig_tr_global[used], rtol=0, atol=1e-13)
    np.testing.assert_allclose(st2.camera_tr_rig, st.camera_tr_rig, atol=1e-13)
    for g2, g in zip(st2.grids, st.grids):
        np.testing.assert_allclose(g2, g, rtol=1e-13, atol=1e-13)
    d = st2.grids[0].reshape(-1, 3)[: pb.cameras[0].grid_points]
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-15)      # re-normalised on load


def test_dataset_to_problem_packs_like_the_host_adapter():
    ds = _example_dataset()
    pb, st = _state()
    mapping = {7: 0, -2: 1, 123456: 2}
    from camera_calibration_amd.problem import Camera, State
    cams = [pb.cameras[0], pb.cameras[0]]
    state = State(st.rig_tr_global[:2], np.tile(st.camera_tr_rig[0], (2, 1)), st.points, [st.grids[0], st.grids[0]])
    p2, s2 = cio.dataset_to_problem(ds, [False, True], cams, state, mapping)
    assert p2.n_images == 1 and p2.n_obs == 1 and p2.obs_camera.tolist() == [1] and p2.obs_image.tolist() == [0]
    assert p2.obs_point.tolist() == [2] and s2.rig_tr_global.shape == (1, 7)


def test_cpp_mirror_reads_and_rewrites_the_files(tmp_path):
    lib_dir = os.path.join(os.path.dirname(os.path.abspath(cio.__file__)))
    host = os.path.join(lib_dir, "libcalib_ba_host_test.so")
    if not os.path.exists(host):
        pytest.skip("host library not built")
    try:
        # libcalib_ba_hip.so (a dependency) must be loadable; the I/O entry points make no HIP calls
        C.CDLL(os.path.join(lib_dir, "libcalib_ba_hip.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(host)
    except OSError as e:
        pytest.skip(f"cannot load host library here: {e}")
    pb, st = _state(0, ncam=2)
    ds = _example_dataset()
    mapping = {7: 0, -2: 1, 123456: 2}
    used = np.array([True, False, True, True, True])
    din, sin_ = str(tmp_path / "in" / "dataset.bin"), str(tmp_path / "in" / "state")
    dout, sout = str(tmp_path / "out" / "dataset.bin"), str(tmp_path / "out" / "state")
    cio.save_dataset(din, ds)
    cio.save_ba_state(sin_, used, pb.cameras, st, mapping)
    ni, nc, npnt, bad = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    nf = C.c_int64(0)
    rc = L.cba_host_io_roundtrip(din.encode(), sin_.encode(), dout.encode(), sout.encode(), C.byref(ni), C.byref(nc), C.byref(nf),
                                 C.byref(npnt), C.byref(bad))
    assert rc == 0
    assert (ni.value, nc.value, nf.value, npnt.value, bad.value) == (2, 2, 4, pb.n_points, 0)
    ds2 = cio.load_dataset(dout)
    assert ds2.image_sizes == ds.image_sizes and [s.filename for s in ds2.imagesets] == [s.filename for s in ds.imagesets]
    for a, b in zip(ds.imagesets, ds2.imagesets):
        for fa, fb in zip(a.features, b.features):
            assert fa.tobytes() == fb.tobytes()
    assert ds2.known_geometries[0].feature_id_to_position == ds.known_geometries[0].feature_id_to_position
    used2, cams2, st2, map2 = cio.load_ba_state(sout)
    np.testing.assert_array_equal(used2, used)
    assert cams2 == pb.cameras and map2 == mapping
    np.testing.assert_allclose(st2.points, st.points, rtol=1e-13)
    np.testing.assert_allclose(st2.rig_tr_global[used], st.rig_tr_global[used], atol=1e-13)
    for g2, g in zip(st2.grids, st.grids):
        np.testing.assert_allclose(g2, g, atol=1e-13)
    # the reference's convenience .obj files next to poses and points (calibration_io.cc:817-836, 923-935): written by both mirrors
    for name in ("rig_tr_global.yaml.obj", "camera_tr_rig.yaml.obj", "points.yaml.obj"):
        a = np.loadtxt(os.path.join(sin_, name), usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
        b = np.loadtxt(os.path.join(sout, name), usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
        assert a.shape == b.shape and a.shape[0] > 0
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    # values that are not re-normalised pass through both writers unchanged: identical text
    assert open(os.path.join(sin_, "points.yaml")).read().split("\n")[1] == open(os.path.join(sout, "points.yaml")).read().split("\n")[1]
