"""Pin-hole camera with optional radial distortion (argument type of the BA entry point).

Own restatement of reference ``src/camera.py`` (PinHoleCamera, camera.py:8-70; ``project``
camera.py:101-165).  ``project`` is the arithmetic the HIP residual kernel restates:
u = fx * x/z + cx, v = fy * y/z + cy, times (1 + k1 r^2 + k2 r^4 + k3 r^6) when any k != 0.
"""

from typing import Union

import numpy as np

from .validation_error import ProcessingError, ValidationError


def _finite_number(v) -> bool:
    return isinstance(v, (int, float)) and bool(np.isfinite(v))


def _check_points(arr, width, name):
    if not isinstance(arr, np.ndarray):
        raise ValidationError(f"{name} must be a numpy array, got {type(arr)}")
    if arr.ndim != 2 or arr.shape[1] != width:
        raise ValidationError(f"{name} must have shape (N, {width}), got {arr.shape}")
    if arr.shape[0] == 0:
        raise ValidationError(f"{name} cannot be empty")
    if not np.all(np.isfinite(arr)):
        raise ValidationError(f"{name} must contain only finite values")


class PinHoleCamera:
    def __init__(self, fx: float, fy: float, cx: float, cy: float,
                 k1: float = 0.0, k2: float = 0.0, k3: float = 0.0):
        for name, val in (("fx", fx), ("fy", fy)):
            if not _finite_number(val) or val <= 0:
                raise ValidationError(f"Focal length {name} must be a positive finite number, got {val}")
        for name, val in (("cx", cx), ("cy", cy)):
            if not _finite_number(val):
                raise ValidationError(f"Principal point {name} must be a finite number, got {val}")
        for name, val in (("k1", k1), ("k2", k2), ("k3", k3)):
            if not _finite_number(val):
                raise ValidationError(f"Radial distortion coefficient {name} must be a finite number, got {val}")
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.k1, self.k2, self.k3 = k1, k2, k3

    # [fx, fy, cx, cy, k1, k2, k3] -- the same 7 scalars the C ABI takes (include/ba_amd.h)
    @classmethod
    def from_vector(cls, vector: np.ndarray) -> "PinHoleCamera":
        return cls(*(vector[i] for i in range(7)))

    def to_vector(self) -> np.ndarray:
        return np.array([self.fx, self.fy, self.cx, self.cy, self.k1, self.k2, self.k3])

    def has_distortion(self) -> bool:
        return self.k1 != 0.0 or self.k2 != 0.0 or self.k3 != 0.0

    def _radial_scale(self, xn: np.ndarray, yn: np.ndarray) -> np.ndarray:
        # reference camera.py:143-156: r_distorted / r, with 1.0 at the optical centre
        r2 = xn**2 + yn**2
        r = np.sqrt(r2)
        factor = 1.0 + self.k1 * r2 + self.k2 * r2**2 + self.k3 * r2**3
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(r == 0, 1.0, (r * factor) / r)

    def project(self, points_3d: np.ndarray) -> np.ndarray:
        """Camera-frame (N, 3) points -> (N, 2) pixels.  Raises ProcessingError when any z == 0;
        points behind the camera are *not* rejected (reference camera.py:124-132)."""
        _check_points(points_3d, 3, "points_3d")
        z = points_3d[:, 2]
        if np.any(z == 0):
            raise ProcessingError("Cannot project points with zero Z coordinates (division by zero)")
        xn = points_3d[:, 0] / z
        yn = points_3d[:, 1] / z
        if self.has_distortion():
            s = self._radial_scale(xn, yn)
            xn = xn * s
            yn = yn * s
        return np.vstack(((self.fx * xn) + self.cx, (self.fy * yn) + self.cy)).T

    def undistort(self, points_2d: np.ndarray) -> np.ndarray:
        """Five fixed-point iterations of x = x_d / scale(x) (reference camera.py:167-236)."""
        _check_points(points_2d, 2, "points_2d")
        if not self.has_distortion():
            return points_2d.copy()
        xd = (points_2d[:, 0] - self.cx) / self.fx
        yd = (points_2d[:, 1] - self.cy) / self.fy
        xn, yn = xd.copy(), yd.copy()
        for _ in range(5):
            s = self._radial_scale(xn, yn)
            xn, yn = xd / s, yd / s
        return np.vstack(((self.fx * xn) + self.cx, (self.fy * yn) + self.cy)).T

    def unproject(self, points_2d: np.ndarray, depths: Union[np.ndarray, float]) -> np.ndarray:
        """Pixels + positive depths -> camera-frame points (reference camera.py:238-308)."""
        _check_points(points_2d, 2, "points_2d")
        n = points_2d.shape[0]
        if isinstance(depths, np.ndarray) and depths.ndim == 0:
            depths = float(depths.item())
        if isinstance(depths, np.ndarray):
            if depths.ndim == 2 and depths.shape == (n, 1):
                depths = depths.flatten()
            elif depths.ndim != 1:
                raise ValidationError(f"depths array must have shape ({n},) or ({n}, 1), got {depths.shape}")
            if depths.shape[0] != n:
                raise ValidationError(f"depths array must have shape ({n},), got {depths.shape}")
            if not np.all(np.isfinite(depths)):
                raise ValidationError("depths must contain only finite values")
            if np.any(depths <= 0):
                raise ValidationError("depths must be positive")
        elif isinstance(depths, (int, float)):
            if not np.isfinite(depths):
                raise ValidationError("depths must be a finite number")
            if depths <= 0:
                raise ValidationError("depths must be positive")
            depths = np.full(n, depths, dtype=float)
        else:
            raise ValidationError(f"depths must be a number or numpy array, got {type(depths)}")
        x = ((points_2d[:, 0] - self.cx) * depths) / self.fx
        y = ((points_2d[:, 1] - self.cy) * depths) / self.fy
        return np.vstack((x, y, depths)).T

    def get_camera_matrix(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])

    def get_distortion_coefficients(self) -> np.ndarray:
        return np.array([self.k1, self.k2, self.k3])

    @classmethod
    def from_camera_matrix(cls, camera_matrix: np.ndarray, k1: float = 0.0, k2: float = 0.0,
                           k3: float = 0.0) -> "PinHoleCamera":
        if not cls.is_valid_camera_matrix(camera_matrix):
            raise ValidationError(f"Invalid camera matrix provided: {camera_matrix}")
        return cls(fx=float(camera_matrix[0, 0]), fy=float(camera_matrix[1, 1]),
                   cx=float(camera_matrix[0, 2]), cy=float(camera_matrix[1, 2]), k1=k1, k2=k2, k3=k3)

    @staticmethod
    def is_valid_camera_matrix(camera_matrix: np.ndarray) -> bool:
        """[[fx,0,cx],[0,fy,cy],[0,0,1]] with fx, fy > 0 (reference camera.py:353-397)."""
        m = camera_matrix
        if not isinstance(m, np.ndarray) or m.shape != (3, 3) or not np.all(np.isfinite(m)):
            return False
        if m[0, 1] != 0.0 or m[1, 0] != 0.0:
            return False
        if not np.allclose(m[2, :], [0.0, 0.0, 1.0]):
            return False
        return bool(m[0, 0] > 0 and m[1, 1] > 0)
