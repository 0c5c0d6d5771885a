"""Batched counterparts of the reference's Monte-Carlo drivers.

random_stress(): scripts/random_stress_test.py:226-486 as ONE projection launch, ONE solver
launch and ONE comparison launch - pose draws in the driver's order (:245-258), projection +
pixel quantisation (PNP_SOLVER_LIB.py:4532-4557), solve_pnp on the six-landmark subset, the
10 cm / 10 deg pass criteria (:374-377, TEST_TOOLBOX.py:55-62) and the landmark error sets of
compare_result_and_generate_result_dict (TEST_TOOLBOX.py:312-343, :453-463) with the failure
counters built on them (:408-415), ground-truth classes (TEST_TOOLBOX.py:133-247), the
per-class error statistics (TEST_TOOLBOX.py:892-937, :1173-1188), their small-/large-angle
restrictions (:959-970) and the four-way class-cell statistics (:975-1064, :1233-1245).

face_variation(): scripts/face_variation_test.py:275-520, :640-735 - the generator's pattern is
perturbed per sample (unit-norm direction over 14 of the 15 landmarks, 2 cm radius) while the
solver keeps the nominal pattern; projection with one pattern per sample, solve and comparison
are three launches; the top decile of each error is then analysed for the most displaced
landmark and the dominant perturbation directions (SVD).

landmark_noise_sweep(): scripts/LM_noise_test.py:141-175, :253-360 - yaw x noise-sigma grid of
noisy landmark sets, all cells solved in one launch, yaw-error mean / stddev / MAE meshes.

The arithmetic of projection and solving runs on the GPU; the statistics are a few NumPy
reductions over the downloaded results (the reference's report layer).
"""
from __future__ import annotations

import numpy as np

from . import batch as _batch
from . import synth

#: "drpy_expand" ground-truth classes (TEST_TOOLBOX.py:137-162): labels and np.digitize bins
DRPY_CLASSES = {
    "depth": ([str(int(v)) for v in np.arange(20.0, 240.1, 20)], list(np.arange(20.0, 240.1, 20)[:-1] + 10.0)),
    "roll": (["-45", "-25", "0", "25", "45"], [-35, -15, 15, 35]),
    "pitch": (["-30", "-15", "0", "15", "30"], [-23, -8, 8, 23]),
    "yaw": (["-40", "-20", "0", "20", "40"], [-30, -10, 10, 30]),
}


def classify(values, class_name):
    """Class label of every value (TEST_TOOLBOX.classify_drpy, :239-247)."""
    labels, bins = DRPY_CLASSES[class_name]
    return np.array(labels)[np.digitize(values, bins)]


STAT_FIELDS = ("n_data", "m_ratio", "mean", "stddev", "max_dev", "MAE_2_GT", "MAE_2_mean")


def error_statistics(est, gt=None, unit_scale=1.0):
    """The reference's statistics of an estimate against ground truth, or of the value itself
    when gt is None (TEST_TOOLBOX.get_statistic_of_result, :892-937)."""
    est = np.asarray(est, dtype=np.float64)
    n = est.shape[0]
    if n == 0:
        return None
    if gt is None:
        err, ratio = est, est
    else:
        gt = np.asarray(gt, dtype=np.float64)
        err, ratio = est - gt, est / gt
    mean = np.average(err)
    return dict(n_data=n, m_ratio=np.average(ratio), mean=mean * unit_scale,
                stddev=(np.sum((err - mean) ** 2) / n) ** 0.5 * unit_scale,
                max_dev=np.max(np.abs(err - mean)) * unit_scale,
                MAE_2_GT=np.sum(np.abs(err)) / n * unit_scale,
                MAE_2_mean=np.sum(np.abs(err - mean)) / n * unit_scale)


def small_angle_mask(roll_gt, pitch_gt, yaw_gt, angle_th=30.0):
    """TEST_TOOLBOX.approval_func_small_angle (:959-967): every ground-truth angle within the
    threshold; its complement is approval_func_large_angle (:969-970)."""
    return (np.abs(roll_gt) <= angle_th) & (np.abs(pitch_gt) <= angle_th) & (np.abs(yaw_gt) <= angle_th)


def class_cell_plan(classes):
    """What class_cell_statistics needs of the classes alone: the four label lists in numeric
    order, the shape of the cell array, and the samples grouped by cell."""
    names = ("distance", "roll", "pitch", "yaw")
    labels, index = [], []
    for k in names:
        uniq, inverse = np.unique(np.asarray(classes[k]), return_inverse=True)
        by_value = sorted(range(len(uniq)), key=lambda i: float(uniq[i]))
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[by_value] = np.arange(len(uniq))
        labels.append(uniq[by_value].tolist())
        index.append(rank[inverse.reshape(-1)])
    shape = tuple(len(lab) for lab in labels)
    flat = np.ravel_multi_index(index, shape)
    order = np.argsort(flat, kind="stable")                 # keeps sample order inside a cell, as the lists do
    cuts = np.flatnonzero(np.diff(flat[order])) + 1
    groups = np.split(order, cuts)
    return labels, shape, [(np.unravel_index(flat[g[0]], shape), g) for g in groups if len(g)]


def class_cell_statistics(classes, est, gt=None, unit_scale=1.0, plan=None):
    """Statistics of one quantity in every (distance, roll, pitch, yaw) class cell
    (TEST_TOOLBOX.get_all_class_seperated_result + get_drpy_statistic, :975-1064).
    classes: dict of four label arrays as returned by random_stress (plan: class_cell_plan(classes),
    when several quantities share the classes).  Returns (labels, A): labels = the four label
    lists in numeric order, A[d, r, p, y, :] = STAT_FIELDS of the cell, NaN where a cell is empty."""
    labels, shape, groups = plan if plan is not None else class_cell_plan(classes)
    est = np.asarray(est, dtype=np.float64)
    gt = None if gt is None else np.asarray(gt, dtype=np.float64)
    A = np.full(shape + (len(STAT_FIELDS),), np.nan)
    for cell, idx in groups:
        st = error_statistics(est[idx], None if gt is None else gt[idx], unit_scale)
        A[cell] = [st[k] for k in STAT_FIELDS]
    return labels, A


def random_stress(count, seed=42, pattern15=None, lm_rows=None, K=synth.K_DEFAULT, is_quantized=True,
                  device="cuda:0", error_bound=(10.0, 10.0, 10.0, 10.0)):
    """Run `count` random-stress samples.  pattern15: (M,3) golden pattern used to generate the
    landmarks (default: the six solver landmarks only); lm_rows: rows of it the solver uses.
    Returns a dict with the ground truth, the estimates, `passed` (count,4) = depth, roll, pitch,
    yaw within bounds, `pass_count`, the normalised landmark errors `lm_err_avg` / `lm_err_max` /
    `lm_err_key` (count,3) for observed-vs-truth, reprojection-vs-observed, reprojection-vs-truth,
    `failed_sample_count`, `failed_sample_fit_error_count`, `stats` = per distance class (and
    "all") statistics of depth (cm), roll, pitch and yaw (deg) errors, and `cell_labels`,
    `cell_stats` = the same statistics (plus those of the normalised observed-vs-truth landmark
    error) in every (distance, roll, pitch, yaw) class cell, see class_cell_statistics."""
    if pattern15 is None:
        pattern15, lm_rows = synth.PATTERN6, list(range(6))
    pattern15 = np.asarray(pattern15, dtype=np.float64)
    rng = np.random.default_rng(seed)
    roll, pitch, yaw, t = synth.sample_stress_poses(rng, count)
    uv_all = _batch.project_batch(roll, yaw, pitch, t, pattern15, K, 1.0 if is_quantized else 0.0, device=device)
    uv = uv_all[:, lm_rows, :].contiguous()
    out = _batch.solve_pnp_batch(uv, pattern15[lm_rows].astype(np.float32), K, method="qeif")
    cmp_ = _batch.compare_batch(out["R"], out["t"], np.stack([roll, yaw, pitch], axis=1), t[:, 2] * 100.0,
                                pattern15, K, uv_obs=uv_all, bound=error_bound)
    res = {k: v.detach().cpu().numpy() for k, v in out.items()}
    res.update({k: v.detach().cpu().numpy() for k, v in cmp_.items() if k != "passed"})
    gt = np.stack([t[:, 2] * 100.0, roll, pitch, yaw], axis=1)
    passed = cmp_["passed"].cpu().numpy()
    failed = passed.sum(axis=1) < 4                                   # pass_count_threshold = 4 (:71, :412)
    fit_failed = failed & (res["lm_err_avg"][:, 1] > 1.0)             # :408, :414
    classes = {"distance": classify(gt[:, 0], "depth"), "roll": classify(roll, "roll"),
               "pitch": classify(pitch, "pitch"), "yaw": classify(yaw, "yaw")}
    stats = {}
    for label in list(dict.fromkeys(classes["distance"].tolist())) + ["all"]:
        m = slice(None) if label == "all" else classes["distance"] == label
        stats[label] = dict(depth=error_statistics(res["t3"][m], t[m, 2], 100.0),
                            roll=error_statistics(res["rpy_deg"][m, 0], roll[m]),
                            pitch=error_statistics(res["rpy_deg"][m, 2], pitch[m]),
                            yaw=error_statistics(res["rpy_deg"][m, 1], yaw[m]))
    cell_stats, plan = {}, class_cell_plan(classes)
    for name, e_, g_, scale in (("depth", res["t3"], t[:, 2], 100.0), ("roll", res["rpy_deg"][:, 0], roll, 1.0),
                                ("pitch", res["rpy_deg"][:, 2], pitch, 1.0), ("yaw", res["rpy_deg"][:, 1], yaw, 1.0),
                                ("LM_GT_error_average_normalize", res["lm_err_avg"][:, 0], None, 1.0)):
        cell_labels, cell_stats[name] = class_cell_statistics(classes, e_, g_, scale, plan)     # TEST_TOOLBOX.py:1240-1245
    res["cell_labels"], res["cell_stats"] = cell_labels, cell_stats
    return dict(roll_gt=roll, pitch_gt=pitch, yaw_gt=yaw, t_gt=t, uv=uv_all.detach().cpu().numpy(), classes=classes,
                passed=passed, pass_count=passed.sum(axis=1), failed_sample_count=int(failed.sum()),
                failed_sample_fit_error_count=int(fit_failed.sum()), stats=stats, **res)


def landmark_noise_sweep(yaw_list, sigma_list, noise, pattern15, lm_rows, K=synth.K_DEFAULT, fixed_depth=1.0,
                         bbox_resolution=112.0, bbox_size=30.0, device="cuda:0"):
    """Yaw x noise-sigma sweep.  noise: (n_noise, M, 2) unit-variance draws (the reference uses
    default_rng(42).multivariate_normal(0, I2, (2000, 15))), scaled by sigma*bbox_size/bbox_resolution
    and added to the clean landmarks, which are quantised with step bbox_size/bbox_resolution.
    Returns yaw_est (n_yaw, n_sigma, n_noise) and the three meshes."""
    import torch
    yaw_list = np.asarray(yaw_list, dtype=np.float64)
    sigma_list = np.asarray(sigma_list, dtype=np.float64)
    pattern15 = np.asarray(pattern15, dtype=np.float64)
    ny, ns, nn = len(yaw_list), len(sigma_list), noise.shape[0]
    q = 1.0 / bbox_resolution * bbox_size
    zeros = np.zeros(ny)
    t = np.tile(np.array([0.0, 0.0, fixed_depth]), (ny, 1))
    clean = _batch.project_batch(zeros, yaw_list, zeros, t, pattern15, K, q, device=None).astype(np.float64)
    # float64 on the host exactly as the driver adds it, then ONE rounding to float32
    scale = (sigma_list / bbox_resolution * bbox_size)[None, :, None, None, None]
    pts = clean[:, None, None, :, :] + scale * np.asarray(noise, dtype=np.float64)[None, None, :, :, :]
    uv = np.ascontiguousarray(pts[:, :, :, lm_rows, :].reshape(ny * ns * nn, len(lm_rows), 2), dtype=np.float32)
    out = _batch.solve_pnp_batch(torch.from_numpy(uv).to(device), pattern15[lm_rows].astype(np.float32), K)
    yaw_est = out["rpy_deg"][:, 1].detach().cpu().numpy().astype(np.float64).reshape(ny, ns, nn)
    err = yaw_est - yaw_list[:, None, None]
    return dict(yaw_est=yaw_est, uv_clean=clean, mesh_mean=err.mean(axis=2), mesh_std=err.std(axis=2),
                mesh_mae=np.abs(err).mean(axis=2))


def draw_pattern_perturbations(rng, count, n_points=15, fixed_row=0, radius=0.02, random_pose=False):
    """The face-variation driver's draws, sample by sample in its order
    (face_variation_test.py:295-308, :351-371): optionally a random-stress pose, then a
    (n_points-1, 3) standard-normal matrix, flattened, normalised to unit length
    (PNP_SOLVER.unit_vec, PNP_SOLVER_LIB.py:4666-4675), scaled by `radius` and read back as
    a (3, n_points-1) matrix whose column j displaces the j-th landmark other than `fixed_row`.
    Returns (roll, pitch, yaw, t (count,3), perturbation (count, n_points, 3))."""
    m = n_points - 1
    roll, pitch, yaw = np.zeros(count), np.zeros(count), np.zeros(count)
    t = np.tile(np.array([0.0, 0.0, 1.0]), (count, 1))                      # :310-323
    pert = np.zeros((count, n_points, 3))
    free = [i for i in range(n_points) if i != fixed_row]
    if random_pose:  # pose and perturbation draws interleave: one sample at a time, as the driver does
        flat = np.empty((count, 3 * m))
        for i in range(count):
            r_, p_, y_, t_ = synth.sample_stress_poses(rng, 1)
            roll[i], pitch[i], yaw[i], t[i] = r_[0], p_[0], y_[0], t_[0]
            flat[i] = rng.multivariate_normal(np.zeros(3), np.eye(3), m).reshape(3 * m)
    else:            # only the perturbations are drawn: one call yields the same stream
        flat = rng.multivariate_normal(np.zeros(3), np.eye(3), (count, m)).reshape(count, 3 * m)
    norm = np.array([np.linalg.norm(f) for f in flat])      # the driver's 1-D norm (a dot product), row by row
    unit = flat * np.where(np.abs(norm) <= 1e-7, 1.0e7, 1.0 / np.maximum(norm, 1e-300))[:, None]
    pert[:, free, :] = np.transpose(radius * unit.reshape(count, 3, m), (0, 2, 1))
    return roll, pitch, yaw, t, pert


def perturbation_analysis(abs_err, perturbation, top_ratio=0.1, k_top_direction=5):
    """Top-decile analysis of one error quantity (face_variation_test.py:640-657 and
    get_most_fragile_point_and_perturbation_direction, :662-735): the samples with the largest
    absolute error (ties in sample order, as the driver's heap pops them), how often each
    landmark carried the largest displacement among them, and the leading right singular
    vectors of their stacked perturbations.  The function in the reference sits in a script that
    cannot be imported, so this restatement has no reference-generated fixture behind it."""
    abs_err = np.asarray(abs_err, dtype=np.float64)
    n_top = int(len(abs_err) * top_ratio)
    order = np.lexsort((np.arange(len(abs_err)), -abs_err))[:n_top]
    top = perturbation[order]                                               # (n_top, M, 3)
    norms = np.linalg.norm(top, axis=2)
    counts = np.bincount(np.argmax(norms, axis=1), minlength=perturbation.shape[1])
    _, sv, vh = np.linalg.svd(top.reshape(n_top, -1))
    k = min(k_top_direction, vh.shape[0])
    return dict(top_idx=order, top_val=abs_err[order], fragile_point_count=counts,
                top_directions=vh[:k].reshape(k, perturbation.shape[1], 3), top_singular_values=sv[:k],
                value_max=float(abs_err[order[0]]) if n_top else float("nan"),
                top_value_mean=float(abs_err[order].mean()) if n_top else float("nan"))


def face_variation(count, seed=42, pattern15=None, fixed_row=0, lm_rows=None, K=synth.K_DEFAULT, is_quantized=True,
                   random_pose=False, radius=0.02, device="cuda:0", error_bound=(10.0, 10.0, 10.0, 10.0)):
    """Run `count` face-variation samples: landmarks generated from a per-sample perturbed copy of
    pattern15 (all rows except fixed_row displaced), solved against the nominal rows lm_rows.
    Returns ground truth, perturbations, pixels, estimates, the comparison epilogue's outputs,
    abs_err (count,4) = depth [m], roll, pitch, yaw [deg], and `analysis` = perturbation_analysis
    per quantity."""
    pattern15 = np.asarray(pattern15, dtype=np.float64)
    rng = np.random.default_rng(seed)
    roll, pitch, yaw, t, pert = draw_pattern_perturbations(rng, count, pattern15.shape[0], fixed_row, radius, random_pose)
    own = pattern15[None, :, :] + pert                                      # the generator's pattern, per sample
    uv_all = _batch.project_batch(roll, yaw, pitch, t, own, K, 1.0 if is_quantized else 0.0, device=device)
    out = _batch.solve_pnp_batch(uv_all[:, lm_rows, :].contiguous(), pattern15[lm_rows].astype(np.float32), K)
    cmp_ = _batch.compare_batch(out["R"], out["t"], np.stack([roll, yaw, pitch], axis=1), t[:, 2] * 100.0,
                                pattern15, K, uv_obs=uv_all, bound=error_bound)
    res = {k: v.detach().cpu().numpy() for k, v in out.items()}
    res.update({k: v.detach().cpu().numpy() for k, v in cmp_.items()})
    abs_err = np.abs(np.stack([res["t3"].astype(np.float64) - t[:, 2], res["rpy_deg"][:, 0] - roll,
                               res["rpy_deg"][:, 2] - pitch, res["rpy_deg"][:, 1] - yaw], axis=1))
    analysis = {q: perturbation_analysis(abs_err[:, c], pert) for c, q in enumerate(("depth", "roll", "pitch", "yaw"))}
    return dict(roll_gt=roll, pitch_gt=pitch, yaw_gt=yaw, t_gt=t, perturbation=pert, uv=uv_all.detach().cpu().numpy(),
                abs_err=abs_err, analysis=analysis, **res)
