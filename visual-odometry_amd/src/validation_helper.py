"""Shape checks shared by the value types (mirrors reference src/validation_helper.py:6-64)."""

import numpy as np

from .validation_error import ValidationError


def _shape_of(a):
    return a.shape if isinstance(a, np.ndarray) else type(a)


class ValidationHelper:
    @staticmethod
    def validate_ids(ids: np.ndarray) -> None:
        # reference validation_helper.py:10-14: TypeError / ValueError, not ValidationError
        if not isinstance(ids, np.ndarray):
            raise TypeError("ids must be a NumPy array.")
        if ids.ndim != 1:
            raise ValueError(f"ids must be 1D, but got {ids.ndim}D.")

    @staticmethod
    def _validate_rows(pts, width, name):
        if not isinstance(pts, np.ndarray) or pts.ndim != 2 or pts.shape[1] != width:
            raise ValidationError(f"{name} must be Nx{width} array, got shape {_shape_of(pts)}")

    @staticmethod
    def validate_pts3d(pts3d: np.ndarray) -> None:
        ValidationHelper._validate_rows(pts3d, 3, "pts3d")

    @staticmethod
    def validate_pts2d(pts2d: np.ndarray) -> None:
        ValidationHelper._validate_rows(pts2d, 2, "pts2d")

    @staticmethod
    def _validate_single(pt, width, name):
        if not isinstance(pt, np.ndarray):
            raise ValidationError(f"{name} must be a numpy array, got {type(pt)}")
        if pt.shape != (width,):
            raise ValidationError(f"{name} must be ({width},) array, got shape {pt.shape}")
        if not np.all(np.isfinite(pt)):
            raise ValidationError(f"{name} must contain only finite values")

    @staticmethod
    def validate_pt3d(pt3d: np.ndarray) -> None:
        ValidationHelper._validate_single(pt3d, 3, "pt3d")

    @staticmethod
    def validate_pt2d(pt2d: np.ndarray) -> None:
        ValidationHelper._validate_single(pt2d, 2, "pt2d")
