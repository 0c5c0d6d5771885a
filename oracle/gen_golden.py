#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference tree is not present on the GPU box):

    python oracle/gen_golden.py [--ref /root/reference/scripts] [--out tests/golden]

The reference modules are imported from where they lie (nothing is copied; bytecode
writing is disabled so the read-only tree is left untouched).  `cv2` is absent in this
image and only needed by the reference's drawing helpers, so an empty stand-in module is
registered for the import of its evaluation toolbox; no cv2 function is ever called.

Each .npz holds inputs and the reference's outputs for one family of cases.  2-D inputs
are float32-representable by construction (integer pixels, or rounded to float32 before
being handed to the reference as float64), so the float32 device path sees the same
numbers.  Seeds and sizes are fixed below; re-running reproduces the files bit for bit
on the same NumPy/LAPACK build.
"""
import argparse
import os
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True

F_CAMERA = 225.68717584155982
K = np.array([[F_CAMERA, 0.0, 160.0], [0.0, F_CAMERA, 120.0], [0.0, 0.0, 1.0]])
LM6 = ['eye_l_96', 'eye_r_97', 'mouse_l_76', 'mouse_r_82', 'nose_t_54', 'chin_t_16']


def load_reference(ref_dir):
    sys.path.insert(0, ref_dir)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    import PNP_SOLVER_LIB as PNPS
    import TEST_TOOLBOX as TTBX
    return PNPS, TTBX


class IterCounter:
    """Counts calls of a bound method; the QEIF loop evaluates its model once per
    iteration, so calls == iterations."""

    def __init__(self, solver, name):
        self.n = 0
        self._orig = getattr(solver, name)
        setattr(solver, name, self)

    def __call__(self, *a, **kw):
        self.n += 1
        return self._orig(*a, **kw)


def draw_stress_pose(rng):
    """random_stress_test.py:245-258, same draw order."""
    ang = 45.0
    roll = rng.uniform(-ang, ang, None)
    pitch = rng.uniform(-ang, ang, None)
    yaw = rng.uniform(-ang, ang, None)
    depth = rng.uniform(20, 225, None) / 100.0
    fov_x = rng.uniform(-45.0, 45.0, None)
    fov_y = rng.uniform(-45.0, 45.0, None)
    t = np.zeros((3, 1))
    t[0, 0] = depth * np.tan(np.deg2rad(fov_x))
    t[1, 0] = depth * np.tan(np.deg2rad(fov_y))
    t[2, 0] = depth
    return roll, pitch, yaw, t


def dict_to_uv(d, keys):
    return np.array([[d[k][0, 0], d[k][1, 0]] for k in keys])


def uv_to_dict(uv, keys):
    return {k: np.array([[uv[i, 0]], [uv[i, 1]], [1.0]]) for i, k in enumerate(keys)}


def pack(results):
    """list of reference 7-tuples -> dict of stacked arrays."""
    return dict(
        R=np.array([r[0] for r in results]),
        t=np.array([r[1][:, 0] for r in results]),
        t3=np.array([r[2] for r in results]),
        roll=np.array([r[3] for r in results]),
        yaw=np.array([r[4] for r in results]),
        pitch=np.array([r[5] for r in results]),
        res_norm=np.array([r[6] for r in results]),
    )


_CELL = {}   # what a worker process of section 15 needs (inherited through fork)


def _noise_cell(task):
    """One (yaw, sigma) cell of the landmark-noise grid: the reference's yaw estimate per noise draw."""
    yi, si = task
    est, keys15 = _CELL['est'], _CELL['keys15']
    clean, s_glob, noise = _CELL['uv_clean'][yi], _CELL['s_glob'][si], _CELL['noise']
    out = np.zeros(len(noise))
    for ni in range(len(noise)):
        pts = (clean + s_glob * noise[ni]).astype(np.float32).astype(np.float64)
        out[ni] = est.solve_pnp(uv_to_dict(pts, keys15))[4]
    return yi, si, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference/scripts')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--n-stress', type=int, default=10000)
    ap.add_argument('--only', default='', help='comma-separated section numbers (1-17); default all')
    ap.add_argument('--workers', type=int, default=8, help='processes of section 15')
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    PNPS, TTBX = load_reference(args.ref)
    t_start = time.time()

    alex = TTBX.get_golden_pattern('Alexander')
    holly = TTBX.get_golden_pattern('Holly')
    keys15 = list(alex.keys())
    gt = PNPS.PNP_SOLVER(K, [alex], pattern_scale_list=[1.0], verbose=False)
    est = PNPS.PNP_SOLVER(K, [alex], pattern_scale_list=[1.0], verbose=False)
    counter = IterCounter(est, 'QEKF_get_hx_H')

    only = set(int(v) for v in args.only.split(',') if v)

    def want(section):
        return not only or section in only

    def save(name, **arrs):
        path = os.path.join(out_dir, name)
        np.savez_compressed(path, **arrs)
        print('wrote %s (%.1f KB)  t=%.0fs' % (name, os.path.getsize(path) / 1024.0, time.time() - t_start))

    # ---- 1. known-answer cases -------------------------------------------------
    if want(1):
        R_known = gt.get_rotation_matrix_from_Euler(15.0, 45.0, -15.0, is_degree=True)
        t_known = np.array([[0.35], [0.35], [0.5]])          # random_stress_test.py:260-270
        img = gt.perspective_projection_golden_landmarks(R_known, t_known, is_quantized=True)
        counter.n = 0
        known = est.solve_pnp(img)
        known_iters = counter.n
        # Euler helpers: round trip of PNP_SOLVER_LIB.py:4727-4732 plus a spread of angles and
        # two gimbal-locked matrices
        e_in, e_R, e_out = [], [], []
        rng = np.random.default_rng(7)
        angles = [(45.0, 40.0, -15.0), (0.0, 0.0, 0.0), (10.0, 35.0, -25.0), (0.0, 0.0, 90.0),
                  (30.0, -20.0, -90.0), (170.0, 100.0, 60.0)]
        angles += [tuple(rng.uniform(-180, 180, 3)) for _ in range(58)]
        for (ro, ya, pi) in angles:
            Rm = gt.get_rotation_matrix_from_Euler(ro, ya, pi, is_degree=True)
            e_in.append([ro, ya, pi])
            e_R.append(Rm)
            e_out.append(gt.get_Euler_from_rotation_matrix(Rm, verbose=False, is_degree=True))
        save('known_cases.npz',
             uv15=dict_to_uv(img, keys15), keys15=np.array(keys15), lm6=np.array(LM6),
             R_gt=R_known, t_gt=t_known[:, 0], iters=np.int32(known_iters),
             euler_in=np.array(e_in), euler_R=np.array(e_R), euler_out=np.array(e_out, dtype=np.float64),
             **{k: v[0] for k, v in pack([known]).items()})

    # ---- 2. random-stress run, QEIF through solve_pnp (15-key dict, 6 used) ------
    if want(2):
        n = args.n_stress
        rng = np.random.default_rng(42)
        draws = np.zeros((n, 3))
        t_gt = np.zeros((n, 3))
        uv15 = np.zeros((n, 15, 2))
        iters = np.zeros(n, dtype=np.int32)
        passed = np.zeros((n, 4), dtype=bool)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt.perspective_projection_golden_landmarks(R, t, is_quantized=True)
            counter.n = 0
            out = est.solve_pnp(img)
            iters[i] = counter.n
            res.append(out)
            draws[i] = [roll, pitch, yaw]
            t_gt[i] = t[:, 0]
            uv15[i] = dict_to_uv(img, keys15)
            flags, _ = TTBX.check_if_the_sample_passed(           # random_stress_test.py:374-377
                (out[2] * 100.0, out[3], out[5], out[4]),
                (t[2, 0] * 100.0, roll, pitch, yaw), (10.0, 10.0, 10.0, 10.0))
            passed[i] = flags
        save('stress_qeif_n6.npz', seed=np.int64(42), roll_gt=draws[:, 0], pitch_gt=draws[:, 1],
             yaw_gt=draws[:, 2], t_gt=t_gt, uv15=uv15.astype(np.float32), keys15=np.array(keys15),
             lm6=np.array(LM6), iters=iters, passed=passed, **pack(res))

    # ---- 3. QEIF on exact (unquantised) projections rounded to float32 -----------
    if want(3):
        n = 2048
        rng = np.random.default_rng(43)
        uv6 = np.zeros((n, 6, 2), dtype=np.float32)
        iters = np.zeros(n, dtype=np.int32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt.perspective_projection_golden_landmarks(R, t, is_quantized=False)
            uv6[i] = dict_to_uv(img, LM6).astype(np.float32)
            counter.n = 0
            res.append(est.solve_pnp(uv_to_dict(uv6[i].astype(np.float64), LM6)))
            iters[i] = counter.n
        save('qeif_n6_exact.npz', seed=np.int64(43), uv=uv6, iters=iters, **pack(res))

    # ---- 4. iterated DLT, 6 and 15 points ----------------------------------------
    if want(4):
        for tag, keys, count, seed in (('dlt_n6', LM6, 2048, 44), ('dlt_n15', keys15, 1024, 45)):
            rng = np.random.default_rng(seed)
            uv = np.zeros((count, len(keys), 2), dtype=np.float32)
            on_axis = np.zeros(count, dtype=bool)
            raised = np.zeros(count, dtype=bool)
            res = []
            pat = est.np_point_3d_pretransfer_dict_list[0]
            for i in range(count):
                roll, pitch, yaw, t = draw_stress_pose(rng)
                if i % 4 == 3:                     # every 4th case on the optical axis, where
                    t[0, 0] = t[1, 0] = 0.0        # this method is accurate
                    on_axis[i] = True
                R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
                img = gt.perspective_projection_golden_landmarks(R, t, is_quantized=True)
                uv[i] = dict_to_uv(img, keys).astype(np.float32)
                try:
                    res.append(est.solve_pnp_single_pattern(uv_to_dict(uv[i].astype(np.float64), keys), pat))
                except np.linalg.LinAlgError:
                    # the reference's inspection SVD (:256-258) raises once an earlier
                    # iteration has produced NaN; recorded as a NaN result
                    raised[i] = True
                    nan = np.float64('nan')
                    res.append((np.full((3, 3), nan), np.full((3, 1), nan), nan, nan, nan, nan, nan))
            save(tag + '.npz', seed=np.int64(seed), uv=uv, keys=np.array(keys), on_axis=on_axis,
                 raised=raised, **pack(res))

    # ---- 5. QEIF with the synthetic 68-point pattern -------------------------------
    if want(5):
        n = 512
        P68 = np.random.default_rng(0).uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (68, 3))
        keys68 = ['p%02d' % i for i in range(68)]
        pat68 = {k: P68[i].tolist() for i, k in enumerate(keys68)}
        gt68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
        est68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
        c68 = IterCounter(est68, 'QEKF_get_hx_H')
        rng = np.random.default_rng(46)
        uv = np.zeros((n, 68, 2), dtype=np.float32)
        iters = np.zeros(n, dtype=np.int32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt68.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt68.perspective_projection_golden_landmarks(R, t, is_quantized=True)
            uv[i] = dict_to_uv(img, keys68).astype(np.float32)
            c68.n = 0
            res.append(est68.solve_pnp_QEIF_single_pattern(
                uv_to_dict(uv[i].astype(np.float64), keys68),
                est68.np_point_3d_pretransfer_dict_list[0], LM_key_list=None))
            iters[i] = c68.n
        save('qeif_n68.npz', seed=np.int64(46), xyz=P68, uv=uv, iters=iters, **pack(res))

    # ---- 6. two golden patterns, min-residual selection -----------------------------
    if want(6):
        n = 512
        est2 = PNPS.PNP_SOLVER(K, [alex, holly], pattern_scale_list=[1.0, 1.0], verbose=False)
        gt2 = PNPS.PNP_SOLVER(K, [alex, holly], pattern_scale_list=[1.0, 1.0], verbose=False)
        rng = np.random.default_rng(47)
        uv = np.zeros((n, 6, 2), dtype=np.float32)
        best = np.zeros(n, dtype=np.int32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt2.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            gt2.set_golden_pattern_id(i % 2)
            img = gt2.perspective_projection_golden_landmarks(R, t, is_quantized=True)
            uv[i] = dict_to_uv(img, LM6).astype(np.float32)
            res.append(est2.solve_pnp(uv_to_dict(uv[i].astype(np.float64), LM6)))
            best[i] = est2.current_golden_pattern_id
        save('multi_pattern.npz', seed=np.int64(47), uv=uv,
             xyz=np.array([[alex[k] for k in LM6], [holly[k] for k in LM6]]),
             true_pattern=(np.arange(n) % 2).astype(np.int32), pattern_id=best, **pack(res))

    # ---- 7. the 12-state Levenberg-Marquardt solver, 6 and 15 points ------------------
    if want(7):
        for tag, keys, count, seed in (('lm12_n6', LM6, 256, 48), ('lm12_n15', keys15, 256, 49)):
            rng = np.random.default_rng(seed)
            sub = {k: alex[k] for k in keys}
            est_lm = PNPS.PNP_SOLVER(K, [sub], verbose=False)
            uv = np.zeros((count, len(keys), 2), dtype=np.float32)
            res = []
            for i in range(count):
                roll, pitch, yaw, t = draw_stress_pose(rng)
                R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
                img = gt.perspective_projection_golden_landmarks(R, t, is_quantized=True)
                uv[i] = dict_to_uv(img, keys).astype(np.float32)
                res.append(est_lm.solve_pnp_LM_single_pattern(
                    uv_to_dict(uv[i].astype(np.float64), keys), est_lm.np_point_3d_pretransfer_dict_list[0]))
            save(tag + '.npz', seed=np.int64(seed), uv=uv, keys=np.array(keys), **pack(res))

    # ---- 8. landmark-noise sweep on a reduced grid (LM_noise_test.py:141-175, :253-360)
    if want(8):
        n_noise_full, n_noise = 2000, 200
        rng = np.random.default_rng(42)
        noise_full = rng.multivariate_normal(np.zeros((2,)), np.eye(2), (n_noise_full, 15))
        noise = noise_full[:n_noise]
        yaw_list = np.array([0.0, 45.0, 90.0 * 13 / 14])
        sigma_list = np.array([0.0, 5.0 * 4 / 9, 5.0])
        q = 1.0 / 112.0 * 30.0
        yaw_est = np.zeros((len(yaw_list), len(sigma_list), n_noise))
        uv_clean = np.zeros((len(yaw_list), 15, 2))
        for yi, yaw_i in enumerate(yaw_list):
            R = gt.get_rotation_matrix_from_Euler(0.0, yaw_i, 0.0, is_degree=True)
            t = np.array([0.0, 0.0, 1.0]).reshape((3, 1))
            lm = gt.perspective_projection_golden_landmarks(R, t, is_quantized=True, quantize_q=q,
                                                            is_returning_homogeneous_vec=False)
            uv_clean[yi] = np.array([lm[k] for k in keys15])
            for si, sigma in enumerate(sigma_list):
                s_glob = sigma / 112.0 * 30.0
                for ni in range(n_noise):
                    # float32-representable inputs (the device path reads float32)
                    pts = (uv_clean[yi] + s_glob * noise[ni]).astype(np.float32).astype(np.float64)
                    out = est.solve_pnp(uv_to_dict(pts, keys15))
                    yaw_est[yi, si, ni] = out[4]
        err = yaw_est - yaw_list[:, None, None]
        save('lm_noise_small.npz', seed=np.int64(42), noise=noise, noise_checksum=np.float64(noise_full.sum()),
             yaw_list=yaw_list, sigma_list=sigma_list, quantize_q=np.float64(q), uv_clean=uv_clean,
             yaw_est=yaw_est, mesh_mean=err.mean(axis=2), mesh_std=err.std(axis=2),
             mesh_mae=np.abs(err).mean(axis=2))

    # ---- 9. classes and per-class statistics of the stress run, by the reference's toolbox ----
    if want(9):
        import contextlib
        import io
        import json
        g = np.load(os.path.join(out_dir, 'stress_qeif_n6.npz'))
        with contextlib.redirect_stdout(io.StringIO()):          # the toolbox prints its tables
            cls = TTBX.get_classification_parameters(drpy_class_format='drpy_expand')
        results = []
        for i in range(len(g['t3'])):
            dist_cm = g['t_gt'][i, 2] * 100.0
            results.append({
                'class': {'distance': TTBX.classify_drpy(cls, dist_cm, class_name='depth'),
                          'roll': TTBX.classify_drpy(cls, g['roll_gt'][i], class_name='roll'),
                          'pitch': TTBX.classify_drpy(cls, g['pitch_gt'][i], class_name='pitch'),
                          'yaw': TTBX.classify_drpy(cls, g['yaw_gt'][i], class_name='yaw')},
                't3_est': g['t3'][i], 'distance_GT': dist_cm * 0.01,   # random_stress_test.py:367
                'roll_est': g['roll'][i], 'roll_GT': g['roll_gt'][i],
                'pitch_est': g['pitch'][i], 'pitch_GT': g['pitch_gt'][i],
                'yaw_est': g['yaw'][i], 'yaw_GT': g['yaw_gt'][i]})
        by_dist = TTBX.get_classified_result(results, class_name='distance', approval_func=None)
        stats = {}
        for label, lst in by_dist.items():                       # TEST_TOOLBOX.py:1184-1188
            kw = dict(class_name='distance', class_label=label, verbose=False)
            stats[label] = {
                'depth': TTBX.get_statistic_of_result(lst, data_est_key='t3_est', data_GT_key='distance_GT', unit='cm', unit_scale=100.0, **kw),
                'roll': TTBX.get_statistic_of_result(lst, data_est_key='roll_est', data_GT_key='roll_GT', unit='deg.', unit_scale=1.0, **kw),
                'pitch': TTBX.get_statistic_of_result(lst, data_est_key='pitch_est', data_GT_key='pitch_GT', unit='deg.', unit_scale=1.0, **kw),
                'yaw': TTBX.get_statistic_of_result(lst, data_est_key='yaw_est', data_GT_key='yaw_GT', unit='deg.', unit_scale=1.0, **kw)}
        doc = {'classes': {k: [r['class'][k] for r in results] for k in ('distance', 'roll', 'pitch', 'yaw')},
               'stats': {lab: {q: {k: float(v) for k, v in d.items()} for q, d in per.items()} for lab, per in stats.items()}}
        path = os.path.join(out_dir, 'stress_stats.json')
        with open(path, 'w') as f:
            json.dump(doc, f)
        print('wrote stress_stats.json (%.1f KB)' % (os.path.getsize(path) / 1024.0))

    # ---- 10. comparison epilogue of the stress driver, by the reference's toolbox -----------
    if want(10):
        import contextlib
        import io
        g = np.load(os.path.join(out_dir, 'stress_qeif_n6.npz'))
        n = len(g['t3'])
        names = ('LM_GT', 'predict_LM', 'predict_GT')
        avg = np.zeros((n, 3))
        mx = np.zeros((n, 3))
        key = np.zeros((n, 3), dtype=np.int32)
        t_gt_est = np.zeros((n, 3))
        failed = fit_failed = 0
        sink = io.StringIO()
        for i in range(n):
            roll_gt, pitch_gt, yaw_gt = g['roll_gt'][i], g['pitch_gt'][i], g['yaw_gt'][i]
            dist_cm = g['t_gt'][i, 2] * 100.0                         # random_stress_test.py:281
            img = uv_to_dict(g['uv15'][i].astype(np.float64), keys15)
            R_est, t_est = g['R'][i], g['t'][i].reshape(3, 1)
            with contextlib.redirect_stdout(sink):
                R_gt = est.get_rotation_matrix_from_Euler(roll_gt, yaw_gt, pitch_gt, is_degree=True)   # :365
                tg = (t_est / g['t3'][i]) * (dist_cm * 0.01)                                           # :367-368
                flags, pass_count = TTBX.check_if_the_sample_passed(
                    (g['t3'][i] * 100.0, g['roll'][i], g['pitch'][i], g['yaw'][i]),
                    (dist_cm, roll_gt, pitch_gt, yaw_gt), (10.0, 10.0, 10.0, 10.0))
                rd, _, _ = TTBX.compare_result_and_generate_result_dict(                               # :387-396
                    est, dict(idx=i, file_name='x', distance=dist_cm, roll=roll_gt, pitch=pitch_gt, yaw=yaw_gt,
                              **{'class': None}),
                    R_est, t_est, (g['roll'][i], g['pitch'][i], g['yaw'][i]), R_gt, tg,
                    (roll_gt, pitch_gt, yaw_gt), g['res_norm'][i], 4 - pass_count, pass_count, flags,
                    np_point_image_dict=img, verbose=False)
            sink.seek(0)
            sink.truncate(0)
            for j, nm in enumerate(names):
                avg[i, j] = rd[nm + '_error_average_normalize']
                mx[i, j] = rd[nm + '_error_max_normalize']
                k = rd[nm + '_error_max_key']
                key[i, j] = keys15.index(k) if k else -1
            t_gt_est[i] = tg[:, 0]
            if pass_count < 4:                                        # :412-415, pass_count_threshold = 4 (:71)
                failed += 1
                if rd['predict_LM_error_average_normalize'] > 1.0:
                    fit_failed += 1
        save('stress_epilogue.npz', avg_normalized=avg, max_normalized=mx, max_key=key, t_gt_est=t_gt_est,
             failed_sample_count=np.int64(failed), failed_sample_fit_error_count=np.int64(fit_failed),
             names=np.array(names))

    # ---- 11. four-way (distance, roll, pitch, yaw) class cells and angle-filtered classes ----
    if want(11):
        import contextlib
        import io
        g = np.load(os.path.join(out_dir, 'stress_qeif_n6.npz'))
        e = np.load(os.path.join(out_dir, 'stress_epilogue.npz'))
        with contextlib.redirect_stdout(io.StringIO()):
            cls = TTBX.get_classification_parameters(drpy_class_format='drpy_expand')
        results = []
        for i in range(len(g['t3'])):
            dist_cm = g['t_gt'][i, 2] * 100.0
            results.append({
                'class': {'distance': TTBX.classify_drpy(cls, dist_cm, class_name='depth'),
                          'roll': TTBX.classify_drpy(cls, g['roll_gt'][i], class_name='roll'),
                          'pitch': TTBX.classify_drpy(cls, g['pitch_gt'][i], class_name='pitch'),
                          'yaw': TTBX.classify_drpy(cls, g['yaw_gt'][i], class_name='yaw')},
                't3_est': g['t3'][i], 'distance_GT': dist_cm * 0.01,
                'roll_est': g['roll'][i], 'roll_GT': g['roll_gt'][i],
                'pitch_est': g['pitch'][i], 'pitch_GT': g['pitch_gt'][i],
                'yaw_est': g['yaw'][i], 'yaw_GT': g['yaw_gt'][i],
                'LM_GT_error_average_normalize': e['avg_normalized'][i, 0]})
        cells, d_l, r_l, p_l, y_l = TTBX.get_all_class_seperated_result(results)          # TEST_TOOLBOX.py:1233
        specs = [('depth', 't3_est', 'distance_GT', 'cm', 100.0), ('roll', 'roll_est', 'roll_GT', 'deg.', 1.0),
                 ('pitch', 'pitch_est', 'pitch_GT', 'deg.', 1.0), ('yaw', 'yaw_est', 'yaw_GT', 'deg.', 1.0),
                 ('LM_GT_error_average_normalize', 'LM_GT_error_average_normalize', None, 'px_m', 1.0)]
        fields = ['n_data', 'm_ratio', 'mean', 'stddev', 'max_dev', 'MAE_2_GT', 'MAE_2_mean']
        arrs = {}
        for name, est_key, gt_key, unit, scale in specs:                                   # :1240-1245
            st = TTBX.get_drpy_statistic(cells, class_name=name, data_est_key=est_key, data_GT_key=gt_key,
                                         unit=unit, unit_scale=scale)
            A = np.full((len(d_l), len(r_l), len(p_l), len(y_l), len(fields)), np.nan)
            for a, d in enumerate(d_l):
                for b, r in enumerate(r_l):
                    for c, pp in enumerate(p_l):
                        for dd, y in enumerate(y_l):
                            s_ = st.get(d, {}).get(r, {}).get(pp, {}).get(y)
                            if s_:
                                A[a, b, c, dd] = [s_['n_data'], s_['m_ratio']] + [s_['%s(%s)' % (k, unit)] for k in fields[2:]]
            arrs['cells_' + name] = A
        # distance classes restricted to small / large ground-truth angles (:959-970, :1174-1175)
        for tag, fn in (('small', TTBX.approval_func_small_angle), ('large', TTBX.approval_func_large_angle)):
            by = TTBX.get_classified_result(results, class_name='distance', include_all=False, approval_func=fn)
            A = np.full((len(d_l), len(fields)), np.nan)
            for a, d in enumerate(d_l):
                s_ = TTBX.get_statistic_of_result(by.get(d, []), data_est_key='yaw_est', data_GT_key='yaw_GT',
                                                  unit='deg.', unit_scale=1.0, verbose=False)
                if s_:
                    A[a] = [s_['n_data'], s_['m_ratio']] + [s_['%s(deg.)' % k] for k in fields[2:]]
            arrs['yaw_by_distance_%s_angles' % tag] = A
        save('stress_cells.npz', d_labels=np.array(d_l), r_labels=np.array(r_l), p_labels=np.array(p_l),
             y_labels=np.array(y_l), fields=np.array(fields), **arrs)

    # ---- 12. face-variation driver: perturbed generator patterns, nominal solver pattern -------
    if want(12):
        import contextlib
        import heapq
        import io
        sink = io.StringIO()
        for tag, n, random_pose in (('fixed_pose', 2000, False), ('random_pose', 600, True)):
            rng = np.random.default_rng(42)
            with contextlib.redirect_stdout(sink):
                gt_solver = PNPS.PNP_SOLVER(K, [alex], pattern_scale_list=[1.0], verbose=False)
                est_solver = PNPS.PNP_SOLVER(K, [alex], pattern_scale_list=[1.0], verbose=False)
            pert = np.zeros((n, 15, 3))
            poses = np.zeros((n, 6))                      # roll, pitch, yaw, t
            uv15 = np.zeros((n, 15, 2))
            abs_err = np.zeros((n, 4))
            res = []
            heaps = {k: [] for k in ('depth', 'roll', 'pitch', 'yaw')}
            for i in range(n):
                with contextlib.redirect_stdout(sink):
                    if random_pose:                                           # face_variation_test.py:295-308
                        roll, pitch, yaw, t = draw_stress_pose(rng)
                    else:                                                     # :310-323
                        roll = pitch = yaw = 0.0
                        t = np.array([[0.0], [0.0], [1.0]])
                    new_dict = {k: list(v) for k, v in alex.items()}
                    M = rng.multivariate_normal(np.zeros((3,)), np.eye(3), 14)   # :351
                    unit = gt_solver.unit_vec(M.reshape((42,)))
                    PT = 0.02 * unit.reshape((3, 14))                          # :359
                    pdict, j = {}, 0
                    for k in new_dict:                                        # :365-371
                        if k == 'eye_c_51':
                            pdict[k] = np.zeros((3, 1))
                        else:
                            pdict[k] = PT[:, j:j + 1]
                            new_dict[k] = list(np.array(new_dict[k]).reshape((3, 1)) + pdict[k])
                            j += 1
                    gt_solver.update_the_selected_golden_pattern(0, new_dict, 1.0)
                    R_gt = gt_solver.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
                    gt_solver.set_golden_pattern_id(0)
                    img = gt_solver.perspective_projection_golden_landmarks(R_gt, t, is_quantized=True)
                    out = est_solver.solve_pnp(img)
                sink.seek(0)
                sink.truncate(0)
                dist_cm = t[2, 0] * 100.0
                abs_err[i] = [abs(out[2] - dist_cm * 0.01), abs(out[3] - roll), abs(out[5] - pitch), abs(out[4] - yaw)]
                for c, q in enumerate(heaps):                                 # :517-520
                    heapq.heappush(heaps[q], (-abs_err[i, c], i))
                pert[i] = np.array([pdict[k][:, 0] for k in keys15])
                poses[i] = [roll, pitch, yaw, t[0, 0], t[1, 0], t[2, 0]]
                uv15[i] = dict_to_uv(img, keys15)
                res.append(out)
            n_top = int(n * 0.1)                                              # :640-641
            tops = {}
            for q in heaps:                                                   # :649-657
                tops['top_idx_' + q] = np.array([heapq.heappop(heaps[q])[1] for _ in range(n_top)], dtype=np.int64)
            save('face_variation_%s.npz' % tag, seed=np.int64(42), keys15=np.array(keys15), perturbation=pert,
                 poses=poses, uv15=uv15.astype(np.float32), abs_err=abs_err, **tops, **pack(res))

    # ---- 13. iterated DLT on planar patterns: the (2N x 8) system is rank-deficient and the
    #          reference's pinv (:293) returns its minimum-norm solution --------------------------
    if want(13):
        import contextlib
        import io
        sink = io.StringIO()
        n_per = 192
        kinds = ('z_zero', 'z_const', 'tilted')
        # coordinates on a grid of 2^-10 m and planes with dyadic coefficients, so that every
        # coordinate is exact in float32 and the float32 device path sees the same planar pattern
        grid = {k: [round(c * 1024.0) / 1024.0 for c in alex[k]] for k in LM6}
        pats = []
        for kind in kinds:
            pat = {}
            for k in LM6:
                x, y, _ = grid[k]
                z = 0.0 if kind == 'z_zero' else (2.0 ** -7 if kind == 'z_const' else 0.25 * x + 0.5 * y)
                pat[k] = [x, y, z]
            pats.append(pat)
        rng = np.random.default_rng(48)
        uv = np.zeros((len(kinds) * n_per, 6, 2), dtype=np.float32)
        xyz = np.zeros((len(kinds) * n_per, 6, 3))
        kind_id = np.zeros(len(kinds) * n_per, dtype=np.int32)
        raised = np.zeros(len(kinds) * n_per, dtype=bool)
        res = []
        for c, pat in enumerate(pats):
            with contextlib.redirect_stdout(sink):
                gtp = PNPS.PNP_SOLVER(K, [pat], verbose=False)
                estp = PNPS.PNP_SOLVER(K, [pat], verbose=False)
            for j in range(n_per):
                i = c * n_per + j
                roll, pitch, yaw, t = draw_stress_pose(rng)
                if j % 2:
                    t[0, 0] = t[1, 0] = 0.0
                R = gtp.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
                img = gtp.perspective_projection_golden_landmarks(R, t, is_quantized=True)
                uv[i] = dict_to_uv(img, LM6).astype(np.float32)
                xyz[i] = np.array([pat[k] for k in LM6])
                kind_id[i] = c
                try:
                    res.append(estp.solve_pnp_single_pattern(uv_to_dict(uv[i].astype(np.float64), LM6),
                                                             estp.np_point_3d_pretransfer_dict_list[0]))
                except np.linalg.LinAlgError:
                    raised[i] = True
                    nan = np.float64('nan')
                    res.append((np.full((3, 3), nan), np.full((3, 1), nan), nan, nan, nan, nan, nan))
        save('dlt_planar.npz', seed=np.int64(48), kinds=np.array(kinds), kind_id=kind_id, uv=uv, xyz=xyz,
             keys=np.array(LM6), raised=raised, **pack(res))

    # ---- 14. QEIF on all 15 landmarks of the golden face, and on a second 68-point pattern ---------
    if want(14):
        import contextlib
        import io
        sink = io.StringIO()
        c15 = IterCounter(est, 'QEKF_get_hx_H')
        n = 1024
        rng = np.random.default_rng(49)
        uv = np.zeros((n, 15, 2), dtype=np.float32)
        iters = np.zeros(n, dtype=np.int32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt.perspective_projection_golden_landmarks(R, t, is_quantized=(i % 2 == 0))
            uv[i] = dict_to_uv(img, keys15).astype(np.float32)
            c15.n = 0
            res.append(est.solve_pnp_QEIF_single_pattern(uv_to_dict(uv[i].astype(np.float64), keys15),
                                                         est.np_point_3d_pretransfer_dict_list[0], LM_key_list=None))
            iters[i] = c15.n
        save('qeif_n15.npz', seed=np.int64(49), keys=np.array(keys15), uv=uv, iters=iters,
             quantized=(np.arange(n) % 2 == 0), **pack(res))

        n = 384
        P68 = np.random.default_rng(1).uniform([-0.09, -0.06, -0.05], [0.09, 0.10, 0.03], (68, 3))
        P68 = np.round(P68 * 4096.0) / 4096.0           # on a 2^-12 m grid: exact in float32
        keys68 = ['q%02d' % i for i in range(68)]
        pat68 = {k: P68[i].tolist() for i, k in enumerate(keys68)}
        with contextlib.redirect_stdout(sink):
            gt68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
            est68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
        c68 = IterCounter(est68, 'QEKF_get_hx_H')
        rng = np.random.default_rng(50)
        uv = np.zeros((n, 68, 2), dtype=np.float32)
        iters = np.zeros(n, dtype=np.int32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt68.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt68.perspective_projection_golden_landmarks(R, t, is_quantized=True, quantize_q=0.25)
            uv[i] = dict_to_uv(img, keys68).astype(np.float32)
            c68.n = 0
            res.append(est68.solve_pnp_QEIF_single_pattern(
                uv_to_dict(uv[i].astype(np.float64), keys68),
                est68.np_point_3d_pretransfer_dict_list[0], LM_key_list=None))
            iters[i] = c68.n
        save('qeif_n68_b.npz', seed=np.int64(50), xyz=P68, uv=uv, iters=iters, **pack(res))

    # ---- 15. the landmark-noise study on its full grid (LM_noise_test.py:141-175, :253-360):
    #          15 yaw angles x 10 noise levels x 2000 draws = 300,000 reference solves, spread over
    #          processes; the per-cell statistics of all draws and every tenth draw's estimate are kept
    if want(15):
        import multiprocessing as mp
        n_noise = 2000
        noise = np.random.default_rng(42).multivariate_normal(np.zeros((2,)), np.eye(2), (n_noise, 15))
        yaw_list = np.linspace(0.0, 90.0, 15)
        sigma_list = np.linspace(0.0, 5.0, 10)
        q = 1.0 / 112.0 * 30.0
        uv_clean = np.zeros((len(yaw_list), 15, 2))
        for yi, yaw_i in enumerate(yaw_list):
            R = gt.get_rotation_matrix_from_Euler(0.0, yaw_i, 0.0, is_degree=True)
            t = np.array([0.0, 0.0, 1.0]).reshape((3, 1))
            lm = gt.perspective_projection_golden_landmarks(R, t, is_quantized=True, quantize_q=q,
                                                            is_returning_homogeneous_vec=False)
            uv_clean[yi] = np.array([lm[k] for k in keys15])
        _CELL.update(est=est, keys15=keys15, uv_clean=uv_clean, s_glob=sigma_list / 112.0 * 30.0, noise=noise)
        yaw_est = np.zeros((len(yaw_list), len(sigma_list), n_noise))
        tasks = [(yi, si) for yi in range(len(yaw_list)) for si in range(len(sigma_list))]
        with mp.get_context('fork').Pool(args.workers) as pool:
            for done, (yi, si, col) in enumerate(pool.imap_unordered(_noise_cell, tasks)):
                yaw_est[yi, si] = col
                if done % 10 == 9:
                    print('  noise grid: %d of %d cells  t=%.0fs' % (done + 1, len(tasks), time.time() - t_start), flush=True)
        err = yaw_est - yaw_list[:, None, None]
        save('lm_noise_full.npz', seed=np.int64(42), noise_checksum=np.float64(noise.sum()), yaw_list=yaw_list,
             sigma_list=sigma_list, quantize_q=np.float64(q), uv_clean=uv_clean, every=np.int64(10),
             yaw_est_sample=yaw_est[:, :, ::10], mesh_mean=err.mean(axis=2), mesh_std=err.std(axis=2),
             mesh_mae=np.abs(err).mean(axis=2))

    # ---- 16. the 12-state LM solver on 68 points (the pattern of section 14's second file)
    if want(16):
        import contextlib
        import io
        n = 128
        P68 = np.random.default_rng(1).uniform([-0.09, -0.06, -0.05], [0.09, 0.10, 0.03], (68, 3))
        P68 = np.round(P68 * 4096.0) / 4096.0           # on a 2^-12 m grid: exact in float32
        keys68 = ['q%02d' % i for i in range(68)]
        pat68 = {k: P68[i].tolist() for i, k in enumerate(keys68)}
        with contextlib.redirect_stdout(io.StringIO()):
            gt68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
            lm68 = PNPS.PNP_SOLVER(K, [pat68], verbose=False)
        rng = np.random.default_rng(51)
        uv = np.zeros((n, 68, 2), dtype=np.float32)
        res = []
        for i in range(n):
            roll, pitch, yaw, t = draw_stress_pose(rng)
            R = gt68.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            img = gt68.perspective_projection_golden_landmarks(R, t, is_quantized=True, quantize_q=0.25)
            uv[i] = dict_to_uv(img, keys68).astype(np.float32)
            res.append(lm68.solve_pnp_LM_single_pattern(uv_to_dict(uv[i].astype(np.float64), keys68),
                                                        lm68.np_point_3d_pretransfer_dict_list[0]))
        save('lm12_n68.npz', seed=np.int64(51), uv=uv, xyz=P68, **pack(res))

    # ---- 17. the pre-transform of the pattern (PNP_SOLVER_LIB.py:27-31, :96-98, :2988-2994), which
    # the reference ships switched off: set on an instance, pattern store rebuilt, then the three
    # live solvers.  Two settings: the constructor's own (R_ah = I, t_ah = (0, 0, -0.5)) and the one
    # its comments name (R_ah = a 45 degree pitch, t_ah = 0).
    if want(17):
        import contextlib
        import io
        n = 48
        rng = np.random.default_rng(61)
        poses = [draw_stress_pose(rng) for _ in range(n)]
        uv = np.zeros((n, 15, 2))
        for i, (roll, pitch, yaw, t) in enumerate(poses):
            R = gt.get_rotation_matrix_from_Euler(roll, yaw, pitch, is_degree=True)
            uv[i] = dict_to_uv(gt.perspective_projection_golden_landmarks(R, t, is_quantized=True), keys15)
        out = {}
        for tag, R_ah, t_ah in (('a', np.eye(3), np.array([[0.0, 0.0, -0.5]]).T),
                                ('b', gt.get_rotation_matrix_from_Euler(0.0, 0.0, 45.0, is_degree=True), np.zeros((3, 1)))):
            with contextlib.redirect_stdout(io.StringIO()):
                pt = PNPS.PNP_SOLVER(K, [alex], pattern_scale_list=[1.0], verbose=False)
            pt.is_using_pre_transform = True
            pt.pre_trans_R_a_h, pt.pre_trans_t_a_h = R_ah, t_ah
            pt.update_the_selected_golden_pattern(0, alex, 1.0)
            moved = pt.np_point_3d_pretransfer_dict_list[0]
            res_q, res_d, res_l, side_R, side_t = [], [], [], [], []
            for i in range(n):
                img = uv_to_dict(uv[i], keys15)
                res_q.append(pt.solve_pnp(img))
                side_R.append(pt.np_R_c_a_est.copy())
                side_t.append(pt.np_t_c_a_est[:, 0].copy())
                with np.errstate(all='ignore'):
                    res_d.append(pt.solve_pnp_single_pattern(img, moved))
                    res_l.append(pt.solve_pnp_LM_single_pattern(img, moved))
            R0 = gt.get_rotation_matrix_from_Euler(10.0, -20.0, 5.0, is_degree=True)
            t0 = np.array([[0.05], [-0.02], [0.9]])
            proj = pt.perspective_projection_golden_landmarks(R0, t0, is_quantized=False, is_pretrans_points=True)
            out.update({tag + '_R_ah': R_ah, tag + '_t_ah': t_ah[:, 0],
                        tag + '_moved': np.array([moved[k][:, 0] for k in keys15]),
                        tag + '_proj_moved': dict_to_uv(proj, keys15),
                        tag + '_side_R': np.array(side_R), tag + '_side_t': np.array(side_t)})
            for name, res in (('qeif', res_q), ('dlt', res_d), ('lm12', res_l)):
                out.update({'%s_%s_%s' % (tag, name, k): v for k, v in pack(res).items()})
        save('pre_transform.npz', seed=np.int64(61), uv15=uv, keys15=np.array(keys15), lm6=np.array(LM6),
             proj_R=R0, proj_t=t0[:, 0], **out)

    print('done in %.0f s; numpy %s' % (time.time() - t_start, np.__version__))


if __name__ == '__main__':
    main()
