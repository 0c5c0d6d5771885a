"""Host-side logic that needs no GPU: the C ABI loads and exports what the header declares,
argument errors are reported before any device work, the drop-in class keeps the reference's
host-visible behaviour, and the rank sharding is a partition."""
import copy
import ctypes
import os
import re

import numpy as np
import pytest

import pnp_solver_test_amd as pnp
from conftest import ROOT, lm6_indices
from oracle import pnp_oracle as po
from pnp_solver_test_amd import _lib

K = po.K_DEFAULT


def header_functions():
    text = open(os.path.join(ROOT, "include", "pnp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pnp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.pnp_version()
    assert lib.pnp_max_points() >= 68


def test_default_params_are_the_reference_constants():
    p = _lib.default_params()
    assert p.q_inv == (225.68 ** 2) / 9.0 and p.r_quat == 0.1 and p.r_delta == 0.01
    assert p.omega0 == 1e-5 and p.max_it == 14 and p.rel_tol == 1e-2 and p.res_old0 == 1e-7
    assert pnp.make_params(max_it=5).max_it == 5
    with pytest.raises(TypeError):
        pnp.make_params(bogus=1)


def test_argument_errors_come_back_as_codes_with_messages():
    lib = _lib.load()
    Kc = _lib.k_array(K)
    buf = (ctypes.c_float * 64)()
    a = ctypes.addressof(buf)
    assert lib.pnp_qeif_batch_f32(a, a, 0, Kc, -1, 6, None, a, a, None, None, None, 0, None) == -1
    assert b"negative" in lib.pnp_last_error()
    assert lib.pnp_qeif_batch_f32(a, a, 0, Kc, 1, 0, None, a, a, None, None, None, 0, None) == -1
    assert lib.pnp_qeif_batch_f32(a, a, 0, Kc, 1, 6, None, a, a, None, None, None, 7, None) == -1
    assert lib.pnp_qeif_batch_f32(None, a, 0, Kc, 1, 6, None, a, a, None, None, None, 0, None) == -1
    assert lib.pnp_qeif_batch_f32(a, a, 5, Kc, 1, 6, None, a, a, None, None, None, 0, None) == -1
    assert lib.pnp_dlt_batch_f32(a, a, 0, Kc, 1, 6, 0, a, a, None, None, 0, None) == -1
    singular = _lib.k_array(np.zeros((3, 3)))
    assert lib.pnp_qeif_batch_f32(a, a, 0, singular, 1, 6, None, a, a, None, None, None, 0, None) == -1
    assert b"singular" in lib.pnp_last_error()
    for bad in (dict(r_quat=0.0), dict(r_delta=-1.0), dict(omega0=0.0), dict(q_inv=float("nan")), dict(max_it=0)):
        prm = pnp.make_params(**bad)
        assert lib.pnp_qeif_batch_f32(a, a, 0, Kc, 1, 6, ctypes.byref(prm), a, a, None, None, None, 0, None) == -1, bad
        assert b"must" in lib.pnp_last_error()
    assert lib.pnp_set_lanes_per_problem(3) == -1
    assert lib.pnp_set_lanes_per_problem(0) == 0
    # the float64 flavour, the generator, the comparison epilogue and the form switch check alike
    assert lib.pnp_qeif_batch_f64(a, a, 0, Kc, 1, pnp._lib.load().pnp_max_points() + 1, None, a, a, None, None, None,
                                  0, None) == -1
    assert lib.pnp_lm12_batch_f64(a, a, 0, Kc, 1, 6, -1.0, 14, a, a, None, None, 0, None) == -1
    assert b"lambda" in lib.pnp_last_error()
    assert lib.pnp_qeif_multi_batch_f32(a, a, 0, Kc, 1, 6, None, a, a, None, None, None, None, 0, None) == -1
    assert b"n_patterns" in lib.pnp_last_error()
    assert lib.pnp_project_batch_f64(a, a, a, 0, Kc, 1, 6, -1.0, a, 0, 0, None) == -1
    assert b"quantize_q" in lib.pnp_last_error()
    assert lib.pnp_compare_batch_f32(a, a, None, a, a, a, Kc, 1, 0, None, 1, a, None, None, None, 0, None) == -1
    assert lib.pnp_compare_batch_f64(a, a, None, a, a, a, Kc, 1, 15, None, 1, None, None, None, None, 0, None) == -1
    assert b"no output" in lib.pnp_last_error()
    assert lib.pnp_set_qeif_form(9) == -1 and lib.pnp_set_qeif_form(_lib.QEIF_FORM_AUTO) == 0
    assert lib.pnp_get_qeif_form(1 << 18, 68) == _lib.QEIF_FORM_MOMENTS
    assert lib.pnp_get_qeif_form(1 << 20, 6) == _lib.QEIF_FORM_POINTS
    # an empty batch is a no-op, not an error, and touches no device
    assert lib.pnp_qeif_batch_f32(None, None, 0, Kc, 0, 6, None, None, None, None, None, None, 0, None) == 0
    assert lib.pnp_euler_deg_f32(None, 0, None, 0, None) == 0


def test_lanes_per_problem_heuristic():
    lib = _lib.load()
    lib.pnp_set_lanes_per_problem(0)
    assert lib.pnp_get_lanes_per_problem(1 << 20, 6, 18) == 1      # a lane per problem
    assert lib.pnp_get_lanes_per_problem(1 << 18, 68, 0) >= 2      # LDS bound for 68 points
    assert lib.pnp_get_lanes_per_problem(1, 68, 0) == 64           # one wavefront per problem
    assert lib.pnp_get_lanes_per_problem(1, 6, 0) >= 2


def test_loading_the_library_starts_no_thread_and_reads_no_environment():
    """The host-pointer pipeline's threads belong to the first call that is large enough to be
    chunked: loading the library alone starts none.  And the library proper carries no getenv: the
    A/B switches live in a second build."""
    import subprocess
    import sys
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "os.environ['PNP_HIP_NO_TORCH'] = '1'\n"
            "from pnp_solver_test_amd import _lib\n"
            "before = len(os.listdir('/proc/self/task'))\n"
            "lib = _lib.load()\n"
            "lib.pnp_version(); lib.pnp_get_lanes_per_problem(1 << 20, 6, 18); lib.pnp_last_kernel_name()\n"
            "print(before, len(os.listdir('/proc/self/task')), lib.pnp_host_lane_threads(0))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    before, after, lanes = (int(v) for v in r.stdout.split())
    assert after == before and lanes == 0
    lib = _lib.load()
    assert lib.pnp_host_lane_threads(0) == 0 and lib.pnp_host_lane_threads(-1) == -1
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in undefined and "hipLaunchKernel" in undefined
    ab = _lib.LIB_PATH.replace("libpnp_hip.so", "libpnp_hip_ab.so")
    if os.path.exists(ab):
        assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", ab], capture_output=True, text=True).stdout
        assert b"A/B" not in lib.pnp_version()


def test_tuning_overrides_belong_to_the_calling_thread():
    """pnp_set_lanes_per_problem / pnp_set_qeif_form change the launches of the thread that called
    them and of no other."""
    import threading
    lib = _lib.load()
    seen = {}

    def other():
        seen["lanes"] = lib.pnp_get_lanes_per_problem(1 << 20, 6, 18)
        seen["form"] = lib.pnp_get_qeif_form(1 << 18, 68)
        lib.pnp_set_lanes_per_problem(8)          # ... and must not leak back either
        seen["own"] = lib.pnp_get_lanes_per_problem(1 << 20, 6, 18)

    lib.pnp_set_lanes_per_problem(16)
    lib.pnp_set_qeif_form(_lib.QEIF_FORM_POINTS)
    try:
        assert lib.pnp_get_lanes_per_problem(1 << 20, 6, 18) == 16
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert seen == dict(lanes=1, form=_lib.QEIF_FORM_MOMENTS, own=8)
        assert lib.pnp_get_lanes_per_problem(1 << 20, 6, 18) == 16
        assert lib.pnp_get_qeif_form(1 << 18, 68) == _lib.QEIF_FORM_POINTS
    finally:
        lib.pnp_set_lanes_per_problem(0)
        lib.pnp_set_qeif_form(_lib.QEIF_FORM_AUTO)
    # a forced lane count may take all of a compute unit's LDS: one lane per problem fits 68 points
    lib.pnp_set_lanes_per_problem(1)
    try:
        assert lib.pnp_get_lanes_per_problem(1 << 18, 68, 0) == 1
        assert lib.pnp_get_lanes_per_problem(1 << 18, 512, 3 * 512) >= 8
    finally:
        lib.pnp_set_lanes_per_problem(0)
    assert _lib.last_launch()["kernel"] == "" or _lib.last_launch()["kernel"].startswith("pnp::")


def test_python_argument_checks():
    with pytest.raises(ValueError):
        pnp.solve_pnp_batch(np.zeros((4, 6, 3), np.float32), np.zeros((6, 3), np.float32), K)
    with pytest.raises(ValueError):
        pnp.solve_pnp_batch(np.zeros((4, 6, 2), np.float32), np.zeros((5, 3), np.float32), K)
    with pytest.raises(ValueError):
        pnp.solve_pnp_batch(np.zeros((4, 6, 2), np.float32), np.zeros((6, 3), np.float32), K, method="lm")


def make_solver(patterns=None, **kw):
    patterns = [po.PATTERN_ALEXANDER] if patterns is None else patterns
    return pnp.PNP_SOLVER(K, patterns, **kw)


def test_constructor_semantics():
    pat = {k: list(v) for k, v in po.PATTERN_ALEXANDER.items()}
    Kin = K.copy()
    s = pnp.PNP_SOLVER(Kin, [pat, pat], pattern_scale_list=[0.5])
    assert s.pattern_scale_list == [0.5, 1.0]                      # padded with 1.0
    pat["eye_l_96"][0] = 9.0
    Kin[0, 0] = 1.0                                                # inputs were deep-copied
    assert s.point_3d_dict_list[0]["eye_l_96"][0] == 0.032 and s.np_K_camera_est[0, 0] == po.F_CAMERA
    assert s.np_point_3d_dict_list[0]["eye_l_96"].shape == (3, 1)
    assert s.np_point_3d_dict_list[0]["eye_l_96"][0, 0] == 0.016
    assert s.current_golden_pattern_id == 0 and s.verbose is False
    assert np.array_equal(s.np_R_c_a_est, np.eye(3)) and s.np_t_c_a_est.shape == (3, 1)
    s2 = copy.deepcopy(s)                                          # face_variation_test.py:933
    s2.update_the_selected_golden_pattern(0, po.PATTERN_HOLLY, 1.0)
    assert s2.np_point_3d_dict_list[0]["eye_l_96"][0, 0] == 0.028
    assert s.np_point_3d_dict_list[0]["eye_l_96"][0, 0] == 0.016


def test_pre_transform_pattern_store_matches_reference(golden):
    """The pre-transform switched on as one would on the reference object (attribute set, pattern
    store rebuilt through update_the_selected_golden_pattern): the moved pattern and its projection
    (is_pretrans_points=True) equal the reference's bit for bit, the plain pattern stays as it was."""
    g = golden("pre_transform")
    keys15 = [str(k) for k in g["keys15"]]
    for tag in ("a", "b"):
        s = make_solver()
        plain_before = {k: v.copy() for k, v in s.np_point_3d_dict_list[0].items()}
        assert all(np.array_equal(s.np_point_3d_pretransfer_dict_list[0][k], plain_before[k]) for k in keys15)
        s.is_using_pre_transform = True
        s.pre_trans_R_a_h, s.pre_trans_t_a_h = g[tag + "_R_ah"], g[tag + "_t_ah"].reshape(3, 1)
        assert s.update_the_selected_golden_pattern(0, po.PATTERN_ALEXANDER, 1.0) is True
        moved = s.np_point_3d_pretransfer_dict_list[0]
        assert list(moved) == keys15 and all(moved[k].shape == (3, 1) for k in keys15)
        assert np.array_equal(np.array([moved[k][:, 0] for k in keys15]), g[tag + "_moved"])
        assert all(np.array_equal(s.np_point_3d_dict_list[0][k], plain_before[k]) for k in keys15)
        proj = s.perspective_projection_golden_landmarks(g["proj_R"], g["proj_t"].reshape(3, 1), is_quantized=False,
                                                         is_pretrans_points=True)
        assert np.array_equal(np.array([proj[k][0:2, 0] for k in keys15]), g[tag + "_proj_moved"])


def test_missing_landmark_raises_keyerror_before_any_device_work():
    s = make_solver()
    img = {k: np.array([[1.0], [2.0], [1.0]]) for k in po.LM6_KEYS[:-1]}
    with pytest.raises(KeyError):
        s.solve_pnp(img)


def test_euler_and_projection_helpers_match_reference(golden):
    g = golden("known_cases")
    s = make_solver()
    for ang, R, out in zip(g["euler_in"], g["euler_R"], g["euler_out"]):
        assert np.array_equal(s.get_rotation_matrix_from_Euler(*ang, is_degree=True), R)
        assert np.array_equal(np.array(s.get_Euler_from_rotation_matrix(R, verbose=False, is_degree=True)), out)
    rad = s.get_Euler_from_rotation_matrix(g["euler_R"][0])
    assert np.allclose(np.rad2deg(rad), g["euler_out"][0])
    img = s.perspective_projection_golden_landmarks(g["R_gt"], g["t_gt"].reshape(3, 1), is_quantized=True)
    assert list(img.keys()) == [str(k) for k in g["keys15"]]
    assert img["eye_l_96"].shape == (3, 1) and img["eye_l_96"][2, 0] == 1.0
    uv = np.array([[img[k][0, 0], img[k][1, 0]] for k in img])
    assert np.array_equal(uv, g["uv15"])
    flat = s.perspective_projection_golden_landmarks(g["R_gt"], g["t_gt"].reshape(3, 1),
                                                     is_returning_homogeneous_vec=False)
    assert flat["eye_l_96"].shape == (2,)
    uv_o, dx, dy, dz = s.perspective_projection_obj_axis(g["R_gt"], g["t_gt"].reshape(3, 1), scale=0.1)
    assert uv_o.shape == dx.shape == dy.shape == dz.shape == (2,)
    assert np.allclose(s.unit_vec(np.array([3.0, 4.0])), [0.6, 0.8])
    assert np.allclose(s.unit_vec(np.zeros(3)), 0.0)


def test_dict_to_array_glue(golden):
    g = golden("known_cases")
    s = make_solver()
    keys15 = [str(k) for k in g["keys15"]]
    img = {k: np.array([[g["uv15"][i, 0]], [g["uv15"][i, 1]], [1.0]]) for i, k in enumerate(keys15)}
    uv = s._image_array(img, pnp.LM_KEY_LIST)
    assert uv.dtype == np.float64 and uv.shape == (1, 6, 2)      # the reference's dtype, unrounded
    assert np.array_equal(uv[0], g["uv15"][lm6_indices(g)])
    # a homogeneous component other than 1 is folded into the pixel: K^-1 [u', v', 1] has the first
    # two components of K^-1 [u, v, w]
    K2 = np.array([[230.0, 1.5, 161.0], [0.0, 226.0, 119.0], [0.0, 0.0, 1.0]])
    s2 = pnp.PNP_SOLVER(K2, [po.PATTERN_ALEXANDER])
    img2 = dict(img)
    img2["eye_r_97"] = np.array([[100.0], [50.0], [-1.0]])
    uv2 = s2._image_array(img2, pnp.LM_KEY_LIST)
    Ki = np.linalg.inv(K2)
    assert np.allclose(Ki[0:2] @ np.array([uv2[0, 1, 0], uv2[0, 1, 1], 1.0]), Ki[0:2] @ np.array([100.0, 50.0, -1.0]),
                       rtol=0, atol=1e-15)
    assert np.array_equal(uv2[0, 0], uv[0, 0]) and np.array_equal(uv2[0, 2:], uv[0, 2:])
    P = s._pattern_array(s.np_point_3d_pretransfer_dict_list[0], pnp.LM_KEY_LIST)
    assert P.dtype == np.float64
    assert np.array_equal(P, po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS))


def test_dropin_module_is_importable_by_its_reference_name():
    import importlib.util
    spec = importlib.util.spec_from_file_location("PNP_SOLVER_LIB", os.path.join(ROOT, "dropin", "PNP_SOLVER_LIB.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.PNP_SOLVER is pnp.PNP_SOLVER
    # every attribute the reference's drivers touch
    for name in ("solve_pnp", "get_rotation_matrix_from_Euler", "get_Euler_from_rotation_matrix",
                 "set_golden_pattern_id", "perspective_projection_golden_landmarks",
                 "perspective_projection_obj_axis", "update_the_selected_golden_pattern", "unit_vec",
                 "solve_pnp_single_pattern", "solve_pnp_QEIF_single_pattern", "lib_print"):
        assert callable(getattr(mod.PNP_SOLVER, name))


def test_shard_bounds_partition():
    for total, world in ((8 << 20, 8), (10, 3), (5, 8), (0, 4), (1 << 20, 1)):
        spans = [pnp.shard_bounds(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert pnp.shard_bounds(8 << 20, 3, 8) == (3 << 20, 4 << 20)
    with pytest.raises(ValueError):
        pnp.shard_bounds(10, 4, 4)


def test_stress_classes_and_statistics_match_reference_toolbox(golden):
    """Classification bins and error statistics of the batched driver against what the
    reference's TEST_TOOLBOX computed for the same 10,000 results."""
    import json
    from pnp_solver_test_amd import stress, synth
    g = golden("stress_qeif_n6")
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "stress_stats.json")))
    dist = stress.classify(g["t_gt"][:, 2] * 100.0, "depth")
    assert dist.tolist() == ref["classes"]["distance"]
    assert stress.classify(g["roll_gt"], "roll").tolist() == ref["classes"]["roll"]
    assert stress.classify(g["pitch_gt"], "pitch").tolist() == ref["classes"]["pitch"]
    assert stress.classify(g["yaw_gt"], "yaw").tolist() == ref["classes"]["yaw"]
    for label, per in ref["stats"].items():
        m = slice(None) if label == "all" else dist == label
        mine = dict(depth=stress.error_statistics(g["t3"][m], g["t_gt"][m, 2], 100.0),
                    roll=stress.error_statistics(g["roll"][m], g["roll_gt"][m]),
                    pitch=stress.error_statistics(g["pitch"][m], g["pitch_gt"][m]),
                    yaw=stress.error_statistics(g["yaw"][m], g["yaw_gt"][m]))
        for q, d in per.items():
            unit = "cm" if q == "depth" else "deg."
            assert mine[q]["n_data"] == d["n_data"]
            assert mine[q]["m_ratio"] == pytest.approx(d["m_ratio"], rel=1e-12)
            for k in ("mean", "stddev", "max_dev", "MAE_2_GT", "MAE_2_mean"):
                assert mine[q][k] == pytest.approx(d["%s(%s)" % (k, unit)], rel=1e-9, abs=1e-12)
    # the vectorised sampler consumes the generator exactly as the per-sample loop does
    roll, pitch, yaw, t = synth.sample_stress_poses(np.random.default_rng(42), 10000)
    assert np.array_equal(roll, g["roll_gt"]) and np.array_equal(t, g["t_gt"])
    uv, _ = synth.stress_batch(10000, 42, pattern=po.pattern_array(po.PATTERN_ALEXANDER))
    assert np.array_equal(uv, g["uv15"])


def test_golden_pattern_tables_match_the_reference(golden):
    from pnp_solver_test_amd import synth
    g = golden("known_cases")
    assert list(synth.get_golden_pattern("Alexander")) == [str(k) for k in g["keys15"]]
    assert synth.get_golden_pattern("Alexander") == po.PATTERN_ALEXANDER
    assert synth.get_golden_pattern("Holly") == po.PATTERN_HOLLY
    assert synth.get_golden_pattern("anything else") == po.PATTERN_ALEXANDER
    assert tuple(synth.LM6_KEYS) == tuple(po.LM6_KEYS) == tuple(str(k) for k in g["lm6"])
    m = golden("multi_pattern")            # the rows the reference's two-pattern run used
    assert np.array_equal(synth.pattern_rows("Alexander", synth.LM6_KEYS), m["xyz"][0])
    assert np.array_equal(synth.pattern_rows("Holly", synth.LM6_KEYS), m["xyz"][1])
    assert np.array_equal(synth.PATTERN6, m["xyz"][0])


def test_header_is_plain_c(tmp_path):
    """include/pnp_hip.h must be consumable from C (the boundary is a C ABI): it compiles as C99
    on its own, and the symbols it declares link against the built library."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "pnp_hip.h"\n'
                   'int main(void) { pnp_qeif_params p; pnp_qeif_default_params(&p);\n'
                   '  return (p.max_it == 14 && pnp_max_points() >= 68 && pnp_version()[0] == \'p\') ? 0 : 1; }\n')
    exe = tmp_path / "use_header"
    libdir = os.path.join(ROOT, "pnp_solver_test_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    str(src), "-L", libdir, "-lpnp_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0       # no device needed for these three calls


def test_class_cell_statistics_against_reference_toolbox(golden):
    """Four-way class cells and the angle-restricted distance classes, computed from the
    reference run's own estimates, against the reference toolbox's numbers."""
    from pnp_solver_test_amd import stress
    g, e, c = golden("stress_qeif_n6"), golden("stress_epilogue"), golden("stress_cells")
    dist_cm = g["t_gt"][:, 2] * 100.0
    classes = {"distance": stress.classify(dist_cm, "depth"), "roll": stress.classify(g["roll_gt"], "roll"),
               "pitch": stress.classify(g["pitch_gt"], "pitch"), "yaw": stress.classify(g["yaw_gt"], "yaw")}
    for name, est, gt, scale in (("depth", g["t3"], dist_cm * 0.01, 100.0), ("roll", g["roll"], g["roll_gt"], 1.0),
                                 ("pitch", g["pitch"], g["pitch_gt"], 1.0), ("yaw", g["yaw"], g["yaw_gt"], 1.0),
                                 ("LM_GT_error_average_normalize", e["avg_normalized"][:, 0], None, 1.0)):
        labels, A = stress.class_cell_statistics(classes, est, gt, scale)
        assert labels == [c[k].tolist() for k in ("d_labels", "r_labels", "p_labels", "y_labels")]
        ref = c["cells_" + name]
        assert np.array_equal(np.isnan(A), np.isnan(ref))
        assert np.nanmax(np.abs(A - ref)) <= 1e-9, name
    small = stress.small_angle_mask(g["roll_gt"], g["pitch_gt"], g["yaw_gt"])
    for tag, mask in (("small", small), ("large", ~small)):
        ref = c["yaw_by_distance_%s_angles" % tag]
        for a, label in enumerate(c["d_labels"].tolist()):
            m = mask & (classes["distance"] == label)
            st = stress.error_statistics(g["yaw"][m], g["yaw_gt"][m])
            assert np.abs(np.array([st[k] for k in stress.STAT_FIELDS]) - ref[a]).max() <= 1e-9, (tag, label)


@pytest.mark.parametrize("tag,random_pose", [("fixed_pose", False), ("random_pose", True)])
def test_face_variation_draws_match_reference_run(golden, tag, random_pose):
    """Pose and pattern-perturbation draws of the face-variation driver, seed 42, against what the
    reference's generator produced; the top-decile ordering against its heap pops."""
    from pnp_solver_test_amd import stress
    g = golden("face_variation_" + tag)
    n = len(g["perturbation"])
    fixed = [str(k) for k in g["keys15"]].index("eye_c_51")
    roll, pitch, yaw, t, pert = stress.draw_pattern_perturbations(np.random.default_rng(42), n, 15, fixed, 0.02,
                                                                  random_pose)
    assert np.array_equal(pert, g["perturbation"])
    assert np.array_equal(np.column_stack([roll, pitch, yaw, t]), g["poses"])
    assert np.allclose(np.linalg.norm(pert.reshape(n, -1), axis=1), 0.02, atol=1e-15)
    for c, q in enumerate(("depth", "roll", "pitch", "yaw")):
        a = stress.perturbation_analysis(g["abs_err"][:, c], pert)
        assert np.array_equal(a["top_idx"], g["top_idx_" + q])
        assert a["fragile_point_count"].sum() == n // 10 and a["fragile_point_count"][fixed] == 0
        assert np.all(np.diff(a["top_singular_values"]) <= 0)
        assert np.allclose(np.linalg.norm(a["top_directions"].reshape(5, -1), axis=1), 1.0)
        assert a["value_max"] == g["abs_err"][:, c].max()


def test_result_arrays_of_an_earlier_call_are_checked_before_reuse():
    """solve_pnp_batch(out=...): an array is taken again only if it has the shape, element type
    and layout the call writes; anything else is refused, not replaced."""
    from pnp_solver_test_amd.batch import _result_array
    fits = lambda a: isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous  # noqa: E731
    fresh = _result_array(None, "R", (4, 3, 3), lambda: np.empty((4, 3, 3), np.float32), fits)
    assert fresh.shape == (4, 3, 3)
    held = dict(R=fresh, t=np.empty((4, 3), np.float64), wide=np.empty((4, 6), np.float32)[:, ::2])
    assert _result_array(held, "R", (4, 3, 3), None, fits) is fresh
    for key, shape in (("R", (5, 3, 3)), ("t", (4, 3)), ("wide", (4, 3)), ("missing", (4,))):
        with pytest.raises(ValueError):
            _result_array(held, key, shape, None, fits)
