"""Drop-in for the reference module `PNP_SOLVER_LIB` (scripts/PNP_SOLVER_LIB.py).

The repository's dropin/ directory holds a top-level module of the same name that re-exports
this one: with dropin/ first on sys.path the reference's drivers
(`import PNP_SOLVER_LIB as PNPS`, random_stress_test.py:9, LM_noise_test.py:9) pick it up
instead of the NumPy original.  `PNP_SOLVER` keeps the constructor, `solve_pnp` and helper signatures
the drivers use; the solvers themselves run in the HIP kernels of libpnp_hip.so through
ctypes.  There is no CPU solver here: without the shared library the import fails.

What runs where
  * solve_pnp, solve_pnp_QEIF_single_pattern, solve_pnp_single_pattern,
    solve_pnp_LM_single_pattern, and the batched
    additions solve_pnp_batch / euler_deg_batch: device kernels.
  * get_rotation_matrix_from_Euler, get_Euler_from_rotation_matrix, perspective_projection*,
    transform_3D_point, unit_vec: host-side NumPy utilities of the synthetic-data generator
    (one 3x3 matrix or one point per call), float64 like the reference.
The dict-based calls (one problem each) hand float64 arrays to the float64 flavour of the
kernels and so agree with the reference to float64 round-off; solve_pnp_batch follows the
dtype of the array it is given (float32 halves the memory traffic of large batches).
"""
from __future__ import annotations

import copy

import numpy as np

from . import _lib
from . import batch as _batch

_lib.load()  # fail at import time, loudly, if the HIP library is missing

#: the fixed landmark subset of solve_pnp (reference :156)
LM_KEY_LIST = ['eye_l_96', 'eye_r_97', 'mouse_l_76', 'mouse_r_82', 'nose_t_54', 'chin_t_16']


class PNP_SOLVER(object):
    """Same public surface as the reference class (scripts/PNP_SOLVER_LIB.py:7-203)."""

    def __init__(self, np_K_camera_est, point_3d_dict_list, pattern_scale_list=None, verbose=False):
        # reference :10-64
        self.verbose = True
        self.np_K_camera_est = copy.deepcopy(np_K_camera_est)
        self.pattern_scale_list = pattern_scale_list
        self.point_3d_dict_list = copy.deepcopy(point_3d_dict_list)
        n_pat = len(self.point_3d_dict_list)
        if self.pattern_scale_list is None:
            self.pattern_scale_list = [1.0 for _ in range(n_pat)]
        elif len(self.pattern_scale_list) < n_pat:
            self.pattern_scale_list += [1.0 for _ in range(n_pat - len(self.pattern_scale_list))]
        self.is_using_pre_transform = False
        self.pre_trans_R_a_h = np.eye(3)
        self.pre_trans_t_a_h = np.array([[0.0, 0.0, -0.5]]).T
        self.np_R_c_a_est = np.eye(3)
        self.np_t_c_a_est = np.zeros((3, 1))
        self.np_point_3d_dict_list = list()
        self.np_point_3d_pretransfer_dict_list = list()
        for pattern, scale in zip(self.point_3d_dict_list, self.pattern_scale_list):
            d, d_pre = self.get_np_point_3d_dict(pattern, scale, self.is_using_pre_transform,
                                                 self.pre_trans_R_a_h, self.pre_trans_t_a_h)
            self.np_point_3d_dict_list.append(d)
            self.np_point_3d_pretransfer_dict_list.append(d_pre)
        self.set_golden_pattern_id(0)
        self.verbose = verbose

    # ---- pattern store (reference :66-135) ----------------------------------------------
    def set_golden_pattern_id(self, id):
        self.current_golden_pattern_id = id

    def get_current_golden_pattern(self):
        return self.np_point_3d_dict_list[self.current_golden_pattern_id]

    def get_current_pretransfered_golden_pattern(self):
        return self.np_point_3d_pretransfer_dict_list[self.current_golden_pattern_id]

    def get_np_point_3d_dict(self, point_3d_dict, pattern_scale, is_using_pre_transform=False,
                             pre_trans_R_a_h=None, pre_trans_t_a_h=None):
        np_dict = dict()
        for k in point_3d_dict:
            np_dict[k] = np.array(point_3d_dict[k]).reshape((3, 1))
            np_dict[k] *= pattern_scale
        np_pre = copy.deepcopy(np_dict)
        if is_using_pre_transform:
            for k in np_pre:
                np_pre[k] = self.transform_3D_point(np_pre[k], pre_trans_R_a_h, pre_trans_t_a_h)
        if self.verbose:
            self.lib_print("-" * 35)
            self.lib_print("3D points in local coordinate:")
            self.lib_print("pattern_scale = %f" % pattern_scale)
            with np.printoptions(suppress=True, precision=4):
                for k in np_dict:
                    self.lib_print("%s:%sp_3D=%s.T | p_3D_pretrans=%s.T"
                                   % (k, " " * (12 - len(k)), str(np_dict[k].T), str(np_pre[k].T)))
            self.lib_print("-" * 35)
        return (np_dict, np_pre)

    def update_the_selected_golden_pattern(self, id, point_3d_dict, pattern_scale):
        self.pattern_scale_list[id] = pattern_scale
        self.point_3d_dict_list[id] = copy.deepcopy(point_3d_dict)
        d, d_pre = self.get_np_point_3d_dict(self.point_3d_dict_list[id], self.pattern_scale_list[id],
                                             self.is_using_pre_transform, self.pre_trans_R_a_h,
                                             self.pre_trans_t_a_h)
        self.np_point_3d_dict_list[id] = d
        self.np_point_3d_pretransfer_dict_list[id] = d_pre
        return True

    def lib_print(self, str=''):
        if self.verbose:
            print(str)

    # ---- dict <-> array glue --------------------------------------------------------------
    @staticmethod
    def _image_array(np_point_image_dict, keys):
        """(1, N, 2) float64 of [u, v] for `keys`, in that order; a missing key raises
        KeyError exactly as the reference's dict lookup does (:3270, :3307)."""
        uv = np.empty((1, len(keys), 2), dtype=np.float64)
        for i, k in enumerate(keys):
            p = np.asarray(np_point_image_dict[k], dtype=np.float64).reshape(-1)
            uv[0, i, 0] = p[0]
            uv[0, i, 1] = p[1]
        return uv

    @staticmethod
    def _pattern_array(np_point_3d_dict, keys):
        return np.array([np.asarray(np_point_3d_dict[k], dtype=np.float64).reshape(3) for k in keys],
                        dtype=np.float64)

    def _finish(self, out):
        """Device result (batch of one) -> the reference's 7-tuple, with its side effects
        (:2986-2987, :3025)."""
        np_R_est = out["R"][0].astype(np.float64)
        np_t_est = out["t"][0].astype(np.float64).reshape((3, 1))
        t3_est = np.float64(out["t"][0, 2])
        roll, yaw, pitch = (np.float64(v) for v in out["rpy_deg"][0])
        res_norm = np.float64(out["res_norm"][0])
        self.np_R_c_a_est = copy.deepcopy(np_R_est)
        self.np_t_c_a_est = copy.deepcopy(np_t_est)
        if self.is_using_pre_transform:                        # reference :2988-2994
            np_R_ch = np_R_est @ self.pre_trans_R_a_h
            np_t_ch = np_R_est @ self.pre_trans_t_a_h + np_t_est
            np_R_est, np_t_est, t3_est = np_R_ch, np_t_ch, np_t_ch[2, 0]
            roll, yaw, pitch = self.get_Euler_from_rotation_matrix(np_R_est, verbose=False, is_degree=True)
        return (np_R_est, np_t_est, t3_est, roll, yaw, pitch, res_norm)

    # ---- solvers ----------------------------------------------------------------------------
    def solve_pnp(self, np_point_image_dict):
        """Reference :144-203: QEIF on the six-landmark subset against every golden pattern,
        keeping the first pattern with the smallest residual.  One kernel launch."""
        keys = LM_KEY_LIST
        self.lib_print("LM_key_list = %s" % str(keys))
        uv = self._image_array(np_point_image_dict, keys)
        patterns = np.stack([self._pattern_array(d, keys) for d in self.np_point_3d_pretransfer_dict_list])
        out = _batch.solve_pnp_batch(uv, None, self.np_K_camera_est, method="qeif", patterns=patterns)
        idx_best = int(out["pattern_id"][0])
        self.lib_print("--> Best fitted pattern is [%d] with res_norm_n_est = %f"
                       % (idx_best, float(out["res_norm"][0])))
        self.set_golden_pattern_id(idx_best)
        return self._finish(out)

    def solve_pnp_QEIF_single_pattern(self, np_point_image_dict, np_point_3d_pretransfer_dict, LM_key_list=None):
        """Reference :2771-3025.  LM_key_list=None uses every key of the pattern dict, in its
        order (:3268-3269, :3300-3301)."""
        keys = list(np_point_3d_pretransfer_dict.keys()) if LM_key_list is None else list(LM_key_list)
        uv = self._image_array(np_point_image_dict, keys)
        xyz = self._pattern_array(np_point_3d_pretransfer_dict, keys)
        out = _batch.solve_pnp_batch(uv, xyz, self.np_K_camera_est, method="qeif")
        return self._finish(out)

    def solve_pnp_single_pattern(self, np_point_image_dict, np_point_3d_pretransfer_dict, LM_key_list=None):
        """Reference :205-430: iterated linear DLT.  As there, LM_key_list is ignored and every
        key of the IMAGE dict is used, in its order (:3070, :3100)."""
        keys = list(np_point_image_dict.keys())
        uv = self._image_array(np_point_image_dict, keys)
        xyz = self._pattern_array(np_point_3d_pretransfer_dict, keys)
        out = _batch.solve_pnp_batch(uv, xyz, self.np_K_camera_est, method="dlt", num_it=3)
        return self._finish(out)

    def solve_pnp_LM_single_pattern(self, np_point_image_dict, np_point_3d_pretransfer_dict, LM_key_list=None):
        """Reference :2567-2769: the 12-state Levenberg-Marquardt solver.  As there, LM_key_list is
        ignored: the pattern contributes all its keys in its own order (:2576) and the image dict
        all of its keys in its own order (:2597)."""
        uv = self._image_array(np_point_image_dict, list(np_point_image_dict.keys()))
        xyz = self._pattern_array(np_point_3d_pretransfer_dict, list(np_point_3d_pretransfer_dict.keys()))
        out = _batch.solve_pnp_batch(uv, xyz, self.np_K_camera_est, method="lm12", num_it=14)
        return self._finish(out)

    def solve_pnp_batch(self, uv, xyz=None, *, method="qeif", params=None, num_it=None, want_rpy=True,
                        all_patterns=False):
        """Batched addition (no reference counterpart): uv (B,N,2); xyz (N,3) or (B,N,3), or
        None for the six-landmark subset of the currently selected golden pattern (N must then
        be 6).  all_patterns=True runs every golden pattern and keeps the best per problem.
        See pnp_solver_test_amd.batch.solve_pnp_batch for the result dict."""
        if all_patterns:
            patterns = np.stack([self._pattern_array(d, LM_KEY_LIST)
                                 for d in self.np_point_3d_pretransfer_dict_list])
            return _batch.solve_pnp_batch(uv, None, self.np_K_camera_est, method=method, params=params,
                                          want_rpy=want_rpy, patterns=patterns)
        if xyz is None:
            xyz = self._pattern_array(self.get_current_pretransfered_golden_pattern(), LM_KEY_LIST)
        return _batch.solve_pnp_batch(uv, xyz, self.np_K_camera_est, method=method, params=params,
                                      num_it=num_it, want_rpy=want_rpy)

    # ---- Euler / projection utilities (host side, reference :4442-4621, :4666-4675) ---------
    def get_rotation_matrix_from_Euler(self, roll, yaw, pitch, is_degree=False):
        yaw = -yaw
        if is_degree:
            roll, yaw, pitch = np.deg2rad(roll), np.deg2rad(yaw), np.deg2rad(pitch)
        c1, s1 = np.cos(roll), np.sin(roll)
        c2, s2 = np.cos(yaw), np.sin(yaw)
        c3, s3 = np.cos(pitch), np.sin(pitch)
        E_roll = np.array([[c1, s1, 0.0], [-s1, c1, 0.0], [0.0, 0.0, 1.0]])
        E_yaw = np.array([[c2, 0.0, -s2], [0.0, 1.0, 0.0], [s2, 0.0, c2]])
        E_pitch = np.array([[1.0, 0.0, 0.0], [0.0, c3, s3], [0.0, -s3, c3]])
        return E_roll @ E_pitch @ E_yaw

    def get_Euler_from_rotation_matrix(self, R_in, verbose=True, is_degree=False):
        eps = 10.0 ** -7
        if np.abs(np.pi / 2.0 - np.arcsin(np.abs(R_in[2, 1]))) <= eps:
            theta_2 = np.sign(-R_in[2, 1]) * (np.pi / 2.0)
            theta_3 = 0.0
            theta_1 = np.arctan2(-R_in[1, 0], R_in[0, 0])
        else:
            theta_1 = np.arctan2(R_in[0, 1], R_in[1, 1])
            theta_3 = np.arctan2(R_in[2, 0], R_in[2, 2])
            c1, c3 = np.cos(theta_1), np.cos(theta_3)
            if np.abs(c1) > np.abs(c3):
                if verbose:
                    self.lib_print("Set c2 as (r22/c1)")
                c2 = R_in[1, 1] / c1
            else:
                if verbose:
                    self.lib_print("Set c2 as (r33/c3)")
                c2 = R_in[2, 2] / c3
            theta_2 = np.arctan2(-R_in[2, 1], c2)
        roll, pitch, yaw = theta_1, theta_2, theta_3
        if is_degree:
            roll, yaw, pitch = np.rad2deg(roll), np.rad2deg(yaw), np.rad2deg(pitch)
        yaw = -yaw
        return (roll, yaw, pitch)

    def transform_3D_point(self, point_3D, R_in, t_in):
        return (R_in @ point_3D + t_in)

    def perspective_projection(self, np_point_3d, np_K_camera, np_R, np_t, is_quantized=False,
                               quantize_q=1.0, is_returning_homogeneous_vec=True):
        _p = np_point_3d.reshape((3, 1))
        _ray = np_K_camera @ (np_R @ _p + np_t)
        projection_no_q = _ray / abs(_ray[2, 0])
        if is_quantized:
            np_point_image = np.around(projection_no_q / quantize_q) * quantize_q
        else:
            np_point_image = projection_no_q
        if is_returning_homogeneous_vec:
            return (np_point_image, projection_no_q)
        return (np_point_image[0:2, 0], projection_no_q[0:2, 0])

    def perspective_projection_obj_axis(self, np_R, np_t, scale=1.0):
        K = self.np_K_camera_est
        kw = dict(is_quantized=False, is_returning_homogeneous_vec=False)
        uv_o, _ = self.perspective_projection(np.array([[0.0, 0.0, 0.0]]), K, np_R, np_t, **kw)
        uv_x1, _ = self.perspective_projection(np.array([[scale, 0.0, 0.0]]), K, np_R, np_t, **kw)
        uv_y1, _ = self.perspective_projection(np.array([[0.0, scale, 0.0]]), K, np_R, np_t, **kw)
        uv_z1, _ = self.perspective_projection(np.array([[0.0, 0.0, scale]]), K, np_R, np_t, **kw)
        return (uv_o, (uv_x1 - uv_o), (uv_y1 - uv_o), (uv_z1 - uv_o))

    def perspective_projection_golden_landmarks(self, np_R, np_t, is_quantized=False, quantize_q=1.0,
                                                is_pretrans_points=False, is_returning_homogeneous_vec=True):
        src = (self.get_current_pretransfered_golden_pattern() if is_pretrans_points
               else self.get_current_golden_pattern())
        np_point_image_dict = dict()
        for k in src:
            img, _ = self.perspective_projection(src[k], self.np_K_camera_est, np_R, np_t,
                                                 is_quantized=is_quantized, quantize_q=quantize_q,
                                                 is_returning_homogeneous_vec=is_returning_homogeneous_vec)
            np_point_image_dict[k] = img
        return np_point_image_dict

    def unit_vec(self, vec_in):
        _norm = np.linalg.norm(vec_in)
        _eps = 10 ** -7
        _norm_inv = 1 / _eps if np.abs(_norm) <= 10 ** -7 else 1.0 / _norm
        return (vec_in * _norm_inv)
