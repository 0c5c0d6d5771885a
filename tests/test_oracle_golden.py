"""The CPU oracles (oracle/pnp_oracle.py, oracle/pnp_oracle.c) against golden vectors
produced by running the reference (tests/golden/*.npz, oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import dlt_comparable, lm6_indices, rot_angle
from oracle import c_oracle, pnp_oracle as po

K = po.K_DEFAULT
P6 = po.pattern_array(po.PATTERN_ALEXANDER, po.LM6_KEYS)
P15 = po.pattern_array(po.PATTERN_ALEXANDER)


def rpy_of(g):
    return np.stack([g["roll"], g["yaw"], g["pitch"]], axis=1)


def test_known_pose_numpy_oracle_is_exact(golden):
    g = golden("known_cases")
    uv6 = g["uv15"][lm6_indices(g)]
    r = po.qeif_solve(uv6, P6, K)
    # same operations through the same numpy.linalg entry points: identical bits
    assert np.array_equal(r["R"], g["R"]) and np.array_equal(r["t"], g["t"])
    assert r["iters"] == int(g["iters"]) == 5
    assert (r["roll"], r["yaw"], r["pitch"]) == (g["roll"], g["yaw"], g["pitch"])
    assert r["res_norm"] == g["res_norm"]
    # the projection that produced the pixels
    uv = po.project(P15, K, g["R_gt"], g["t_gt"], is_quantized=True)
    assert np.array_equal(uv, g["uv15"])


def test_euler_helpers(golden):
    g = golden("known_cases")
    for ang, R, out in zip(g["euler_in"], g["euler_R"], g["euler_out"]):
        assert np.array_equal(po.euler_to_R(*ang, is_degree=True), R)
        assert np.array_equal(np.array(po.R_to_euler(R, is_degree=True)), out)
    assert np.abs(c_oracle.euler_deg(g["euler_R"]) - g["euler_out"]).max() < 1e-11
    # reference main(): (45, 40, -15) deg survives the round trip (PNP_SOLVER_LIB.py:4727-4732)
    assert np.allclose(g["euler_out"][0], [45.0, 40.0, -15.0], atol=1e-12)


def test_gimbal_lock_branch():
    R = po.euler_to_R(20.0, 0.0, 90.0, is_degree=True)   # pitch = 90 deg -> |R21| = 1
    assert abs(abs(R[2, 1]) - 1.0) < 1e-15
    roll, yaw, pitch = po.R_to_euler(R, is_degree=True)
    assert pitch == pytest.approx(90.0) and yaw == 0.0
    assert np.allclose(c_oracle.euler_deg(R[None])[0], [roll, yaw, pitch], atol=1e-12)


def test_stress_pose_sampler_matches_reference_draws(golden):
    g = golden("stress_qeif_n6")
    roll, pitch, yaw, t = po.sample_random_stress_poses(np.random.default_rng(int(g["seed"])), 500)
    assert np.array_equal(roll, g["roll_gt"][:500]) and np.array_equal(pitch, g["pitch_gt"][:500])
    assert np.array_equal(yaw, g["yaw_gt"][:500]) and np.array_equal(t, g["t_gt"][:500])
    for i in (0, 17, 499):
        R = po.euler_to_R(roll[i], yaw[i], pitch[i], is_degree=True)
        assert np.array_equal(po.project(P15, K, R, t[i], True).astype(np.float32), g["uv15"][i])


def test_qeif_n6_quantised(golden):
    g = golden("stress_qeif_n6")
    uv = g["uv15"][:, lm6_indices(g), :].astype(np.float64)
    r = c_oracle.qeif_batch(uv, P6, K)
    assert np.array_equal(r["iters"], g["iters"])
    assert np.abs(r["R"] - g["R"]).max() < 1e-10
    assert np.abs(r["t"] - g["t"]).max() < 1e-10
    assert np.abs(r["res_norm"] - g["res_norm"]).max() < 1e-11
    assert np.abs(r["rpy"] - rpy_of(g)).max() < 1e-8
    for i in range(0, 10000, 250):       # the NumPy restatement is operation-for-operation
        q = po.qeif_solve(uv[i], P6, K)
        assert np.array_equal(q["R"], g["R"][i]) and q["iters"] == g["iters"][i]
    # statistics the survey quotes: 4..8 iterations, median 5
    assert int(np.median(g["iters"])) == 5 and g["iters"].max() <= 8


def test_qeif_n6_exact_inputs(golden):
    g = golden("qeif_n6_exact")
    r = c_oracle.qeif_batch(g["uv"].astype(np.float64), P6, K)
    assert np.array_equal(r["iters"], g["iters"])
    assert np.abs(r["R"] - g["R"]).max() < 1e-10 and np.abs(r["t"] - g["t"]).max() < 1e-10


def test_qeif_n68(golden):
    g = golden("qeif_n68")
    assert np.array_equal(g["xyz"], po.synthetic_face68())
    r = c_oracle.qeif_batch(g["uv"].astype(np.float64), g["xyz"], K)
    assert np.array_equal(r["iters"], g["iters"])
    assert np.abs(r["R"] - g["R"]).max() < 1e-10 and np.abs(r["t"] - g["t"]).max() < 1e-10
    q = po.qeif_solve(g["uv"][3].astype(np.float64), g["xyz"], K)
    assert np.array_equal(q["R"], g["R"][3])


def test_qeif_n15_and_a_second_68_point_pattern(golden):
    """All 15 landmarks of the golden face (half the problems with unquantised pixels), and a
    68-point pattern on a 2^-12 m grid with quarter-pixel quantisation, as the reference solved
    them."""
    g = golden("qeif_n15")
    r = c_oracle.qeif_batch(g["uv"].astype(np.float64), P15, K)
    q = g["quantized"]
    assert np.array_equal(r["iters"][q], g["iters"][q])
    same = r["iters"] == g["iters"]
    assert same.mean() >= 0.97          # exact pixels stop at the rounding floor: see the N = 6 case
    assert np.abs(r["R"] - g["R"])[same].max() < 1e-10 and np.abs(r["t"] - g["t"])[same].max() < 1e-10
    assert np.abs(r["res_norm"] - g["res_norm"])[same].max() < 1e-11
    for i in (0, 1, 500):
        o = po.qeif_solve(g["uv"][i].astype(np.float64), P15, K)
        assert np.array_equal(o["R"], g["R"][i]) and o["iters"] == g["iters"][i]
    b = golden("qeif_n68_b")
    assert np.array_equal(b["xyz"], b["xyz"].astype(np.float32))
    r = c_oracle.qeif_batch(b["uv"].astype(np.float64), b["xyz"], K)
    assert np.array_equal(r["iters"], b["iters"])
    assert np.abs(r["R"] - b["R"]).max() < 1e-10 and np.abs(r["t"] - b["t"]).max() < 1e-10
    assert np.abs(r["rpy"] - rpy_of(b)).max() < 1e-8


def test_numpy_batched_restatement(golden):
    """The array-operation form of the NumPy oracle (the bench's second CPU baseline) against the
    reference's results: same iteration counts, poses to round-off, both pattern layouts."""
    g = golden("stress_qeif_n6")
    uv = g["uv15"][:3000, lm6_indices(g), :].astype(np.float64)
    r = po.qeif_solve_batch(uv, P6, K)
    assert np.array_equal(r["iters"], g["iters"][:3000])
    assert np.abs(r["R"] - g["R"][:3000]).max() < 1e-10 and np.abs(r["t"] - g["t"][:3000]).max() < 1e-10
    assert np.abs(r["res_norm"] - g["res_norm"][:3000]).max() < 1e-11
    own = np.broadcast_to(P6, (3000, 6, 3))
    assert np.array_equal(po.qeif_solve_batch(uv, own, K)["R"], r["R"])
    b = golden("qeif_n68_b")
    r = po.qeif_solve_batch(b["uv"].astype(np.float64), b["xyz"], K)
    assert np.array_equal(r["iters"], b["iters"]) and np.abs(r["R"] - b["R"]).max() < 1e-10
    five = po.qeif_solve_batch(uv[:50], P6, K, max_it=5, rel_tol=0.0)
    assert np.all(five["iters"] == 5)
    for i in (0, 7, 49):
        assert np.abs(five["R"][i] - po.qeif_solve(uv[i], P6, K, max_it=5, rel_tol=0.0)["R"]).max() < 1e-11


def test_qeif_per_problem_patterns_equal_shared(golden):
    g = golden("stress_qeif_n6")
    uv = g["uv15"][:64, lm6_indices(g), :].astype(np.float64)
    a = c_oracle.qeif_batch(uv, P6, K)
    b = c_oracle.qeif_batch(uv, np.broadcast_to(P6, (64, 6, 3)).copy(), K)
    assert np.array_equal(a["R"], b["R"])


@pytest.mark.parametrize("tag,P", [("dlt_n6", P6), ("dlt_n15", P15)])
def test_dlt(golden, tag, P):
    g = golden(tag)
    uv = g["uv"].astype(np.float64)
    r = c_oracle.dlt_batch(uv, P, K)
    ok = ~g["raised"] & np.isfinite(g["R"]).all(axis=(1, 2))
    # the QR-based C oracle agrees with the reference's SVD-based pinv wherever both are finite;
    # one side alone is non-finite only where k2 is an exact zero up to round-off
    away = dlt_comparable(r["R"], g["R"], r["margins"])
    assert away.sum() >= len(away) - 16
    assert np.abs(r["R"] - g["R"])[away].max() < 1e-9
    assert np.abs(r["t"] - g["t"])[away].max() < 1e-9
    assert np.abs(r["res_norm"] - g["res_norm"])[away].max() < 1e-9
    for i in np.flatnonzero(ok)[:40]:
        q = po.dlt_solve(uv[i], P, K)
        assert np.abs(q["R"] - g["R"][i]).max() < 1e-12 and abs(q["res_norm"] - g["res_norm"][i]) < 1e-12
    # on the optical axis the method is accurate (rotation within a few degrees of truth)
    assert np.median(g["res_norm"][ok & g["on_axis"]]) < np.median(g["res_norm"][ok & ~g["on_axis"]])


def test_dlt_planar_patterns(golden):
    """Rank-deficient systems: the NumPy oracle calls the same pinv as the reference (:293)."""
    g = golden("dlt_planar")
    assert not g["raised"].any() and np.isfinite(g["R"]).all()
    import itertools
    z0 = list(g["kinds"]).index("z_zero")
    forced = 0
    for i in range(0, len(g["uv"]), 3):
        uv_i = g["uv"][i].astype(np.float64)
        q = po.dlt_solve(uv_i, g["xyz"][i], K)
        if g["kind_id"][i] == z0 and np.abs(q["R"] - g["R"][i]).max() >= 1e-11:
            # z = 0 plane: the similarity that picks the mirror solution of a round is an exact
            # zero, its computed sign round-off; the reference run is one of the 2^3 sign sequences
            assert q["min_abs_sim"] <= 1e-15
            q = min((po.dlt_solve(uv_i, g["xyz"][i], K, se_override=seq) for seq in itertools.product((1, -1), repeat=3)),
                    key=lambda c: np.abs(c["R"] - g["R"][i]).max())
            forced += 1
        assert np.abs(q["R"] - g["R"][i]).max() < 1e-10 and np.abs(q["t"] - g["t"][i]).max() < 1e-10
        assert abs(q["res_norm"] - g["res_norm"][i]) < 1e-10
    assert forced <= 6
    # every coordinate of these patterns is exact in float32, so the float32 device path sees them
    assert np.array_equal(g["xyz"].astype(np.float32).astype(np.float64), g["xyz"])


def test_multi_pattern_selection(golden):
    g = golden("multi_pattern")
    uv = g["uv"].astype(np.float64)
    r0 = c_oracle.qeif_batch(uv, g["xyz"][0], K)
    r1 = c_oracle.qeif_batch(uv, g["xyz"][1], K)
    pid = (r1["res_norm"] < r0["res_norm"]).astype(np.int32)
    assert np.array_equal(pid, g["pattern_id"])
    Rb = np.where(pid[:, None, None] == 1, r1["R"], r0["R"])
    assert np.abs(Rb - g["R"]).max() < 1e-10
    q = po.solve_pnp_multi(uv[5], [g["xyz"][0], g["xyz"][1]], K)
    assert q["pattern_id"] == g["pattern_id"][5] and np.array_equal(q["R"], g["R"][5])


@pytest.mark.parametrize("tag,keys", [("lm12_n6", po.LM6_KEYS), ("lm12_n15", None)])
def test_lm12(golden, tag, keys):
    g = golden(tag)
    P = po.pattern_array(po.PATTERN_ALEXANDER, keys)
    for i in range(0, 256, 16):
        q = po.lm12_solve(g["uv"][i].astype(np.float64), P, K)
        assert np.array_equal(q["R"], g["R"][i]) and np.array_equal(q["t"], g["t"][i])
        assert q["res_norm"] == g["res_norm"][i]


def test_lm12_on_68_points(golden):
    g = golden("lm12_n68")
    for i in range(0, 128, 8):
        q = po.lm12_solve(g["uv"][i].astype(np.float64), g["xyz"], K)
        assert np.array_equal(q["R"], g["R"][i]) and np.array_equal(q["t"], g["t"][i])
        assert q["res_norm"] == g["res_norm"][i]


def test_landmark_noise_sweep(golden):
    g = golden("lm_noise_small")
    noise = np.random.default_rng(int(g["seed"])).multivariate_normal(np.zeros(2), np.eye(2), (2000, 15))
    assert noise.sum() == g["noise_checksum"] and np.array_equal(noise[:200], g["noise"])
    idx = [list(po.PATTERN_ALEXANDER).index(k) for k in po.LM6_KEYS]
    yaw_est = np.zeros_like(g["yaw_est"])
    for yi, yaw in enumerate(g["yaw_list"]):
        R = po.euler_to_R(0.0, yaw, 0.0, is_degree=True)
        clean = po.project(P15, K, R, np.array([0.0, 0.0, 1.0]), True, float(g["quantize_q"]))
        assert np.array_equal(clean, g["uv_clean"][yi])
        for si, sigma in enumerate(g["sigma_list"]):
            pts = (clean[None] + (sigma / 112.0 * 30.0) * g["noise"]).astype(np.float32).astype(np.float64)
            yaw_est[yi, si] = c_oracle.qeif_batch(pts[:, idx, :], P6, K)["rpy"][:, 1]
    assert np.abs(yaw_est - g["yaw_est"]).max() < 1e-7
    err = yaw_est - g["yaw_list"][:, None, None]
    assert np.abs(err.mean(axis=2) - g["mesh_mean"]).max() < 1e-8
    assert np.abs(err.std(axis=2) - g["mesh_std"]).max() < 1e-8


def test_landmark_noise_sweep_full_grid(golden):
    """The study's own grid, 15 yaw x 10 sigma x 2000 draws, all 300,000 solves run by the reference:
    the C oracle against every tenth estimate and against the per-cell statistics of all draws."""
    g = golden("lm_noise_full")
    noise = np.random.default_rng(int(g["seed"])).multivariate_normal(np.zeros(2), np.eye(2), (2000, 15))
    assert noise.sum() == g["noise_checksum"]
    idx = [list(po.PATTERN_ALEXANDER).index(k) for k in po.LM6_KEYS]
    yaw_est = np.zeros((15, 10, 2000))
    for yi, yaw in enumerate(g["yaw_list"]):
        R = po.euler_to_R(0.0, yaw, 0.0, is_degree=True)
        clean = po.project(P15, K, R, np.array([0.0, 0.0, 1.0]), True, float(g["quantize_q"]))
        assert np.array_equal(clean, g["uv_clean"][yi])
        for si, sigma in enumerate(g["sigma_list"]):
            pts = (clean[None] + (sigma / 112.0 * 30.0) * noise).astype(np.float32).astype(np.float64)
            yaw_est[yi, si] = c_oracle.qeif_batch(pts[:, idx, :], P6, K)["rpy"][:, 1]
    assert np.abs(yaw_est[:, :, ::int(g["every"])] - g["yaw_est_sample"]).max() < 1e-7
    err = yaw_est - g["yaw_list"][:, None, None]
    assert np.abs(err.mean(axis=2) - g["mesh_mean"]).max() < 1e-8
    assert np.abs(err.std(axis=2) - g["mesh_std"]).max() < 1e-8
    assert np.abs(np.abs(err).mean(axis=2) - g["mesh_mae"]).max() < 1e-8


def test_compare_epilogue_against_reference_toolbox(golden):
    """The drivers' comparison epilogue against TEST_TOOLBOX.compare_result_and_generate_result_dict
    run on the reference's own stress results."""
    g, e = golden("stress_qeif_n6"), golden("stress_epilogue")
    P15 = po.pattern_array(po.PATTERN_ALEXANDER)
    failed = fit_failed = 0
    for i in range(0, 10000, 7):
        r = po.compare_epilogue(g["R"][i], g["t"][i], (g["roll"][i], g["yaw"][i], g["pitch"][i]),
                                (g["roll_gt"][i], g["yaw_gt"][i], g["pitch_gt"][i]), g["t_gt"][i, 2] * 100.0,
                                g["uv15"][i], P15)
        assert np.abs(r["avg"] - e["avg_normalized"][i]).max() <= 1e-12
        assert np.abs(r["max"] - e["max_normalized"][i]).max() <= 1e-12
        assert r["key"].tolist() == e["max_key"][i].tolist()
        assert np.abs(r["t_gt_est"] - e["t_gt_est"][i]).max() <= 1e-15
        assert r["passed"].tolist() == g["passed"][i].tolist()
    assert int(e["failed_sample_count"]) == int((g["passed"].sum(axis=1) < 4).sum())
