"""ctypes binding of libba_amd.so (include/ba_amd.h) -- the thin shim between the Python entry
point and the HIP kernels.  There is no CPU fallback: if the library or a GPU is missing, every call
raises (the product path must fail loudly rather than silently compute elsewhere)."""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .validation_error import ProcessingError, ValidationError

BA_HISTORY = 128
_LIB_PATH = os.environ.get("BA_AMD_LIB") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libba_amd.so")

STATUS_NAMES = {0: "max_iters", 1: "gtol", 2: "ftol", 3: "xtol"}
# scipy.optimize.least_squares' names for its `loss=` argument -> ba_loss (include/ba_amd.h)
LOSSES = {"linear": 0, "soft_l1": 1, "huber": 2, "cauchy": 3, "arctan": 4}


class BaOptions(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32), ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
        ("lambda0", C.c_double), ("lambda_min", C.c_double), ("lambda_max", C.c_double),
        ("min_diag", C.c_double), ("accept_ratio", C.c_double), ("guard_cheirality", C.c_int32),
        ("linear_solver", C.c_int32), ("pcg_tol", C.c_double), ("pcg_max_iter", C.c_int32),
        ("pcg_check_every", C.c_int32), ("lm_check_every", C.c_int32), ("device", C.c_int32), ("verbose", C.c_int32),
        ("profile_kernels", C.c_int32), ("loss", C.c_int32), ("f_scale", C.c_double),
    ]


class BaReport(C.Structure):
    _fields_ = [
        ("cost0", C.c_double), ("cost", C.c_double), ("grad_inf", C.c_double), ("lambda_", C.c_double),
        ("rms_reproj", C.c_double), ("iterations", C.c_int32), ("accepted", C.c_int32),
        ("linearizations", C.c_int32), ("status", C.c_int32), ("pcg_iters", C.c_int32), ("nobs", C.c_int32),
        ("bad_pivots", C.c_int32), ("guard_rejections", C.c_int32), ("shortened_steps", C.c_int32),
        ("reserved_", C.c_int32),
        ("solve_ms", C.c_double), ("setup_ms", C.c_double), ("n_history", C.c_int32),
        ("cost_history", C.c_double * BA_HISTORY), ("lambda_history", C.c_double * BA_HISTORY),
        ("pcg_history", C.c_int32 * BA_HISTORY), ("accepted_history", C.c_int32 * BA_HISTORY),
    ]

    def as_dict(self) -> dict:
        n = self.n_history
        return dict(cost0=self.cost0, cost=self.cost, grad_inf=self.grad_inf, lam=self.lambda_,
                    rms_reproj=self.rms_reproj, iterations=self.iterations, accepted=self.accepted,
                    linearizations=self.linearizations, status=self.status,
                    status_name=STATUS_NAMES.get(self.status, "?"), pcg_iters=self.pcg_iters, nobs=self.nobs,
                    bad_pivots=self.bad_pivots, guard_rejections=self.guard_rejections,
                    shortened_steps=self.shortened_steps,
                    solve_ms=self.solve_ms, setup_ms=self.setup_ms, cost_history=list(self.cost_history[:n]),
                    lambda_history=list(self.lambda_history[:n]), pcg_history=list(self.pcg_history[:n]),
                    accepted_history=list(self.accepted_history[:n]))


_lib = None

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)
_SIGNATURES = {
    "ba_version": (C.c_int, []),
    "ba_device_count": (C.c_int, []),
    "ba_default_options": (None, [C.POINTER(BaOptions)]),
    "ba_create": (C.c_int, [C.POINTER(BaOptions), C.POINTER(_P)]),
    "ba_destroy": (None, [_P]),
    "ba_last_error": (C.c_char_p, [_P]),
    "ba_set_problem": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _I, _I, _D, _D, C.POINTER(C.c_uint8)]),
    "ba_set_pose_prior": (C.c_int, [_P, C.c_int32, _I, _D, _D, _D, C.c_double]),
    "ba_solve": (C.c_int, [_P, _D, _D, C.POINTER(BaReport)]),
    "ba_upload_guess": (C.c_int, [_P, _D, _D]),
    "ba_solve_resident": (C.c_int, [_P, C.POINTER(BaReport)]),
    "ba_download_estimate": (C.c_int, [_P, _D, _D]),
    "ba_solve_batched": (C.c_int, [_P, C.c_int32, _I, _I, _I, _I, _I, _D, _D, _D, _D, _D, _D, C.POINTER(BaReport)]),
    "ba_evaluate": (C.c_int, [_P, _D, _D, _D, _D, _I]),
    "ba_evaluate_static": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _I, _I, _D, _D, _D, _D, _D, _D, _I,
                                     C.c_char_p, C.c_int32]),
    "ba_triangulate": (C.c_int, [_P, _D, C.c_int32, C.c_double, _D, _D, C.POINTER(C.c_uint8)]),
    "ba_jacobian_blocks": (C.c_int, [_P, _D, _D, _D, _D]),
    "ba_debug_linearization": (C.c_int, [_P, _D, _D, C.c_double] + [_D] * 9),
    "ba_reduced_solver": (C.c_int, [_P, _I]),
    "ba_schur_block_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_problem_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "ba_get_kernel_times": (C.c_int, [_P, _D, _I]),
    "ba_set_kernel_timing": (C.c_int, [_P, C.c_int32]),
    "ba_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ba_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_char_p]),
    "ba_comm_info": (C.c_int, [_P, _I, _I]),
    "ba_debug_table_digests": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int32, _I]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """Load libba_amd.so (does not touch the GPU).  Raises RuntimeError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `make -C visual-odometry_amd/csrc` "
                               "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def comm_unique_id() -> bytes:
    """128-byte RCCL unique id; rank 0 creates it and ships it to the other ranks."""
    buf = C.create_string_buffer(128)
    rc = load_library().ba_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"ba_comm_unique_id failed with status {rc} (librccl.so not loadable?)")
    return buf.raw


def _dptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_D)


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValidationError(f"expected array of shape {shape}, got {a.shape}")
    return a


STATIC_MAX_OBS, STATIC_MAX_CAMS, STATIC_MAX_PTS = 131072, 1024, 65536  # BA_STATIC_MAX_* (include/ba_amd.h)


def fits_static(n_cams: int, n_pts: int, n_obs: int) -> bool:
    return 0 < n_cams <= STATIC_MAX_CAMS and 0 < n_pts <= STATIC_MAX_PTS and 0 < n_obs <= STATIC_MAX_OBS


def evaluate_static(n_cams, n_pts, obs_cam, obs_pt, obs_uv, intr, poses, points, want_residuals=True, device=-1):
    """ba_evaluate_static: residuals, cost and behind-camera count without a handle and without any
    allocation on the device (static scratch inside the library; sizes up to STATIC_MAX_*)."""
    lib = load_library()
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
    n = len(obs_cam)
    if obs_pt.shape != (n,):
        raise ValidationError("obs_cam and obs_pt must have the same length")
    obs_uv = _f64(obs_uv, (n, 2))
    intr = _f64(intr, (7,))
    poses = _f64(poses, (n_cams, 7))
    points = _f64(points, (n_pts, 3))
    res = np.empty(2 * n) if want_residuals else None
    cost = C.c_double()
    nb = C.c_int32()
    err = C.create_string_buffer(512)
    rc = lib.ba_evaluate_static(int(device), int(n_cams), int(n_pts), n, obs_cam.ctypes.data_as(_I),
                                obs_pt.ctypes.data_as(_I), _dptr(obs_uv), _dptr(intr), _dptr(poses), _dptr(points),
                                _dptr(res), C.byref(cost), C.byref(nb), err, 512)
    if rc != 0:
        msg = err.value.decode("utf-8", "replace")
        if rc == 1:
            raise ValidationError(msg)
        if rc == 2:
            raise ProcessingError(msg)
        raise RuntimeError(f"libba_amd status {rc}: {msg}")
    return res, cost.value, nb.value


class Solver:
    """One GPU-resident BA problem (wraps a ba_handle).  Use as a context manager or call close()."""

    def __init__(self, **options):
        lib = load_library()
        self._lib = lib
        self.opt = BaOptions()
        lib.ba_default_options(C.byref(self.opt))
        for k, v in options.items():
            if not hasattr(self.opt, k):
                raise ValidationError(f"unknown solver option {k!r}")
            if k == "loss" and isinstance(v, str):
                if v not in LOSSES:
                    raise ValidationError(f"`loss` must be one of {sorted(LOSSES)}, got {v!r}")
                v = LOSSES[v]
            setattr(self.opt, k, v)
        self._h = _P()
        rc = lib.ba_create(C.byref(self.opt), C.byref(self._h))
        if rc != 0:
            self._h = None
            if rc == 1:
                raise ValidationError("ba_create rejected the options (loss must be a ba_loss value, f_scale > 0, "
                                      "device an existing ordinal)")
            raise RuntimeError(f"ba_create failed with status {rc}: no usable MI355X / HIP device "
                               "(the solver has no CPU fallback)")
        self.C = self.L = self.n = 0

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ba_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc == 0:
            return
        msg = self._lib.ba_last_error(self._h).decode("utf-8", "replace")
        if rc == 1:
            raise ValidationError(msg)
        if rc == 2:
            raise ProcessingError(msg)
        raise RuntimeError(f"libba_amd status {rc}: {msg}")

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Join the landmark-sharded multi-GPU solve (call before set_problem)."""
        if len(unique_id) != 128:
            raise ValidationError("unique_id must be the 128 bytes of comm_unique_id()")
        self._check(self._lib.ba_comm_init(self._h, rank, world, unique_id))

    def table_digests(self) -> list:
        """Digests of the tables of the last set_problem (ba_debug_table_digests; needs BA_TABLE_DIGEST=1 in the environment)."""
        n = C.c_int32()
        self._check(self._lib.ba_debug_table_digests(self._h, None, 0, C.byref(n)))
        buf = (C.c_uint64 * max(1, n.value))()
        self._check(self._lib.ba_debug_table_digests(self._h, buf, n.value, C.byref(n)))
        return [int(v) for v in buf[: n.value]]

    def comm_info(self):
        """(rank, world) as the communicator itself reports them (ba_comm_info); (0, 1) without one."""
        r, w = C.c_int32(), C.c_int32()
        self._check(self._lib.ba_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    # -- problem
    def set_problem(self, n_cams, n_pts, obs_cam, obs_pt, obs_uv, intr, fixed=None):
        obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
        obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
        n = len(obs_cam)
        if obs_pt.shape != (n,):
            raise ValidationError("obs_cam and obs_pt must have the same length")
        obs_uv = _f64(obs_uv, (n, 2))
        intr = _f64(intr, (7,))
        fx = None
        if fixed is not None:
            fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
            if fixed.shape != (n_cams,):
                raise ValidationError("fixed mask must have one entry per camera")
            fx = fixed.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self._lib.ba_set_problem(self._h, n_cams, n_pts, n, obs_cam.ctypes.data_as(_I),
                                             obs_pt.ctypes.data_as(_I), _dptr(obs_uv), _dptr(intr), fx))
        self.C, self.L, self.n = int(n_cams), int(n_pts), int(n)

    PRIOR_MAX_CAMS = 16

    def set_pose_prior(self, cams, ref_poses, H, g, c0: float = 0.0):
        """Quadratic prior on the poses `cams` of the NEXT set_problem (ba_set_pose_prior; cams=None or empty removes
        it): E = c0 + g^T d + 0.5 d^T H d over the chart coordinates d of those poses around ref_poses."""
        n = 0 if cams is None else len(cams)
        if n == 0:
            self._check(self._lib.ba_set_pose_prior(self._h, 0, None, None, None, None, 0.0))
            return
        cams = np.ascontiguousarray(cams, dtype=np.int32)
        ref = _f64(ref_poses, (n, 7))
        H = _f64(H, (6 * n, 6 * n))
        g = _f64(g, (6 * n,))
        self._check(self._lib.ba_set_pose_prior(self._h, n, cams.ctypes.data_as(_I), _dptr(ref), _dptr(H), _dptr(g),
                                                float(c0)))

    def _state(self, poses, points):
        return _f64(poses, (self.C, 7)).copy(), _f64(points, (self.L, 3)).copy()

    def solve(self, poses, points):
        poses, points = self._state(poses, points)
        rep = BaReport()
        self._check(self._lib.ba_solve(self._h, _dptr(poses), _dptr(points), C.byref(rep)))
        return poses, points, rep.as_dict()

    def upload_guess(self, poses, points):
        poses, points = self._state(poses, points)
        self._check(self._lib.ba_upload_guess(self._h, _dptr(poses), _dptr(points)))

    def solve_resident(self) -> dict:
        rep = BaReport()
        self._check(self._lib.ba_solve_resident(self._h, C.byref(rep)))
        return rep.as_dict()

    def solve_batched(self, problems, intr):
        """Batched local bundle adjustment (ba_solve_batched): `problems` is a sequence of dicts with
        obs_cam / obs_pt (indices local to the problem), obs_uv, poses (Cb x 7, row 0 fixed) and points
        (Lb x 3).  Returns (list of poses, list of points, cost0[B], cost[B], report)."""
        B = len(problems)
        if B == 0:
            raise ValidationError("solve_batched needs at least one problem")
        cam_off = np.zeros(B + 1, np.int32)
        lm_off = np.zeros(B + 1, np.int32)
        obs_off = np.zeros(B + 1, np.int32)
        for b, p in enumerate(problems):
            cam_off[b + 1] = cam_off[b] + len(p["poses"])
            lm_off[b + 1] = lm_off[b] + len(p["points"])
            obs_off[b + 1] = obs_off[b] + len(p["obs_cam"])
        obs_cam = np.ascontiguousarray(np.concatenate([np.asarray(p["obs_cam"]) for p in problems]), dtype=np.int32)
        obs_pt = np.ascontiguousarray(np.concatenate([np.asarray(p["obs_pt"]) for p in problems]), dtype=np.int32)
        obs_uv = _f64(np.concatenate([np.asarray(p["obs_uv"], dtype=np.float64).reshape(-1, 2) for p in problems]))
        poses = _f64(np.concatenate([_f64(p["poses"]).reshape(-1, 7) for p in problems])).copy()
        points = _f64(np.concatenate([_f64(p["points"]).reshape(-1, 3) for p in problems])).copy()
        if len(obs_pt) != len(obs_cam) or len(obs_uv) != len(obs_cam):
            raise ValidationError("obs_cam, obs_pt and obs_uv must have the same length in every problem")
        intr = _f64(intr, (7,))
        cost0 = np.empty(B)
        cost = np.empty(B)
        rep = BaReport()
        ip = lambda a: a.ctypes.data_as(_I)
        self._check(self._lib.ba_solve_batched(self._h, B, ip(cam_off), ip(lm_off), ip(obs_off), ip(obs_cam), ip(obs_pt),
                                               _dptr(obs_uv), _dptr(intr), _dptr(poses), _dptr(points), _dptr(cost0),
                                               _dptr(cost), C.byref(rep)))
        self.C, self.L, self.n = int(cam_off[B]), int(lm_off[B]), int(obs_off[B])
        return ([poses[cam_off[b]:cam_off[b + 1]] for b in range(B)], [points[lm_off[b]:lm_off[b + 1]] for b in range(B)],
                cost0, cost, rep.as_dict())

    def download_estimate(self, poses, points):
        poses, points = self._state(poses, points)
        self._check(self._lib.ba_download_estimate(self._h, _dptr(poses), _dptr(points)))
        return poses, points

    def evaluate(self, poses, points, want_residuals=True):
        poses, points = self._state(poses, points)
        res = np.empty(2 * self.n) if want_residuals else None
        cost = C.c_double()
        nb = C.c_int32()
        self._check(self._lib.ba_evaluate(self._h, _dptr(poses), _dptr(points), _dptr(res), C.byref(cost),
                                          C.byref(nb)))
        return res, cost.value, nb.value

    def triangulate(self, poses, points=None, refine_iters: int = 0, min_depth: float = 0.0):
        """Triangulate every landmark of the problem from the given poses (ba_triangulate).  Returns
        (points L x 3, reprojection error L, valid L); rows of `points` (zeros if not given) that are not
        valid are left as passed."""
        poses = _f64(poses, (self.C, 7)).copy()
        points = np.zeros((self.L, 3)) if points is None else _f64(points, (self.L, 3)).copy()
        err = np.empty(self.L)
        valid = np.zeros(self.L, dtype=np.uint8)
        self._check(self._lib.ba_triangulate(self._h, _dptr(poses), int(refine_iters), float(min_depth), _dptr(points),
                                             _dptr(err), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return points, err, valid.astype(bool)

    def jacobian_blocks(self, poses, points):
        poses, points = self._state(poses, points)
        Jc = np.empty((self.n, 2, 6))
        Jp = np.empty((self.n, 2, 3))
        self._check(self._lib.ba_jacobian_blocks(self._h, _dptr(poses), _dptr(points), _dptr(Jc), _dptr(Jp)))
        return Jc, Jp

    def debug_linearization(self, poses, points, lam, want_S=True):
        poses, points = self._state(poses, points)
        Cn, L, n = self.C, self.L, self.n
        out = dict(U=np.empty((Cn, 6, 6)), V=np.empty((L, 3, 3)), W=np.empty((n, 6, 3)), gc=np.empty((Cn, 6)),
                   gp=np.empty((L, 3)), S=np.empty((6 * Cn, 6 * Cn)) if want_S else None, rhs=np.empty(6 * Cn),
                   dc=np.empty((Cn, 6)), dp=np.empty((L, 3)))
        self._check(self._lib.ba_debug_linearization(
            self._h, _dptr(poses), _dptr(points), float(lam), *[_dptr(out[k]) for k in
                                                                ("U", "V", "W", "gc", "gp", "S", "rhs", "dc", "dp")]))
        return out

    KERNEL_CLASSES = ("k_linearize", "k_schur", "k_backsub", "k_cost", "pcg", "other", "k_gather_S", "k_lm_decide")

    def kernel_times(self) -> dict:
        ms = np.zeros(8)
        cnt = np.zeros(8, dtype=np.int32)
        self._check(self._lib.ba_get_kernel_times(self._h, _dptr(ms), cnt.ctypes.data_as(_I)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    def set_kernel_timing(self, on: bool):
        self._check(self._lib.ba_set_kernel_timing(self._h, 1 if on else 0))

    REDUCED_SOLVERS = ("pcg", "pcg_one_workgroup", "dense_cholesky", "banded_cholesky", "component_cholesky",
                       "cyclic_reduction")

    def reduced_solver(self) -> str:
        v = C.c_int32()
        self._check(self._lib.ba_reduced_solver(self._h, C.byref(v)))
        return self.REDUCED_SOLVERS[v.value]

    INFO_KEYS = ("s_blocks", "contrib_rows", "schur_chunks", "u_slabs", "reduced_solver", "band_bw", "band_arrow",
                 "band_segments", "free_cameras", "observed_landmarks", "linearize_flat", "components", "u_rows")

    def problem_info(self) -> dict:
        v = (C.c_int64 * len(self.INFO_KEYS))()
        self._check(self._lib.ba_problem_info(self._h, v))
        return {k: int(v[i]) for i, k in enumerate(self.INFO_KEYS)}

    def schur_block_count(self) -> int:
        v = C.c_int64()
        self._check(self._lib.ba_schur_block_count(self._h, C.byref(v)))
        return v.value
