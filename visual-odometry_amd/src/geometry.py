"""Quaternion / rotation helpers needed on the bundle-adjustment path.

Own restatement of the subset of reference ``src/geometry.py`` that the hot path and its scene
builders use.  Conventions are the reference's: quaternions are
``[w, x, y, z]``, ``R(q)`` rotates camera -> world (geometry.py:192-213), Euler angles are
``[roll, pitch, yaw]`` composed as ``Rz @ Ry @ Rx`` (geometry.py:366-390).  The arithmetic order of
every formula follows the reference so that scenes rebuilt here are bit-identical to scenes built
with the reference (checked by tests/test_scenes_golden.py against tests/golden/).
"""

from typing import Tuple

import numpy as np

from .validation_error import ProcessingError, ValidationError

_UNIT_TOL = 1e-6


def _need(arr, shape, name):
    if not isinstance(arr, np.ndarray) or arr.shape != shape:
        got = arr.shape if isinstance(arr, np.ndarray) else type(arr)
        raise ValidationError(f"{name} must be a numpy array of shape {shape}, got {got}")


def _need_finite(arr, name):
    if not np.all(np.isfinite(arr)):
        raise ValidationError(f"{name} must contain only finite values")


class GeometryUtils:
    """Static helpers; same names and argument meaning as the reference class."""

    # ------------------------------------------------------------------ vectors
    @staticmethod
    def normalize_vector(vector: np.ndarray) -> np.ndarray:
        """Unit-length copy of a (3,) vector or of each row of an (N, 3) array (geometry.py:12-50)."""
        if not isinstance(vector, np.ndarray):
            raise ValidationError(f"vector must be a numpy array, got {type(vector)}")
        if vector.ndim == 1:
            if vector.shape[0] != 3:
                raise ValidationError(f"Vector must have 3 elements, got {vector.shape[0]}")
            length = np.linalg.norm(vector)
            if length == 0:
                raise ProcessingError("Cannot normalize zero-length vector")
            return vector / length
        if vector.ndim == 2:
            if vector.shape[1] != 3:
                raise ValidationError(f"Vectors must have shape (N, 3), got {vector.shape}")
            lengths = np.linalg.norm(vector, axis=1)
            if np.any(lengths == 0):
                raise ProcessingError("Cannot normalize zero-length vectors")
            return vector / lengths[:, np.newaxis]
        raise ValidationError(f"Vector must be 1D or 2D, got {vector.ndim}D")

    # -------------------------------------------------------------- quaternions
    @staticmethod
    def validate_quaternion(quaternion: np.ndarray, atol: float = _UNIT_TOL) -> bool:
        """True for a finite (4,) array with | ||q|| - 1 | <= atol (geometry.py:622-639)."""
        if not isinstance(quaternion, np.ndarray) or quaternion.shape != (4,):
            return False
        if not np.all(np.isfinite(quaternion)):
            return False
        return bool(abs(np.linalg.norm(quaternion) - 1.0) <= atol)

    @staticmethod
    def quaternion_multiply(q1: np.ndarray, q2: np.ndarray) -> np.ndarray:
        """Hamilton product q1 (x) q2 (geometry.py:482-512)."""
        _need(q1, (4,), "q1")
        _need(q2, (4,), "q2")
        if not np.all(np.isfinite(q1)) or not np.all(np.isfinite(q2)):
            raise ValidationError("quaternions must contain only finite values")
        a0, a1, a2, a3 = q1
        b0, b1, b2, b3 = q2
        return np.array(
            [
                a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
                a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0,
            ]
        )

    @staticmethod
    def quaternion_conjugate(quaternion: np.ndarray) -> np.ndarray:
        _need(quaternion, (4,), "quaternion")
        return quaternion * np.array([1.0, -1.0, -1.0, -1.0])

    @staticmethod
    def quaternion_inverse(quaternion: np.ndarray) -> np.ndarray:
        """conj(q) / ||q||^2 (geometry.py:535-553)."""
        _need(quaternion, (4,), "quaternion")
        n2 = np.sum(quaternion**2)
        if n2 == 0:
            raise ProcessingError("Cannot invert zero quaternion")
        return GeometryUtils.quaternion_conjugate(quaternion) / n2

    @staticmethod
    def quaternion_from_axis_angle(axis: np.ndarray, angle: float) -> np.ndarray:
        """[cos(a/2), axis * sin(a/2)] with the axis normalised first (geometry.py:100-124)."""
        _need(axis, (3,), "axis")
        if not np.isfinite(angle):
            raise ValidationError(f"angle must be finite, got {angle}")
        unit = GeometryUtils.normalize_vector(axis)
        half = angle / 2.0
        v = unit * np.sin(half)
        return np.array([np.cos(half), v[0], v[1], v[2]])

    # ------------------------------------------------------------------ matrices
    @staticmethod
    def rotation_matrix_from_quaternion(quaternion: np.ndarray) -> np.ndarray:
        """R(q); q is re-normalised only when its norm is off by more than 1e-6 (geometry.py:185-213)."""
        _need(quaternion, (4,), "quaternion")
        _need_finite(quaternion, "quaternion")
        n = np.linalg.norm(quaternion)
        if n == 0:
            raise ProcessingError("Cannot convert zero quaternion to rotation matrix")
        if abs(n - 1.0) > _UNIT_TOL:
            quaternion = quaternion / n
        w, x, y, z = quaternion
        R = np.empty((3, 3))
        R[0, 0] = 1 - 2 * y * y - 2 * z * z
        R[0, 1] = 2 * x * y - 2 * w * z
        R[0, 2] = 2 * x * z + 2 * w * y
        R[1, 0] = 2 * x * y + 2 * w * z
        R[1, 1] = 1 - 2 * x * x - 2 * z * z
        R[1, 2] = 2 * y * z - 2 * w * x
        R[2, 0] = 2 * x * z - 2 * w * y
        R[2, 1] = 2 * y * z + 2 * w * x
        R[2, 2] = 1 - 2 * x * x - 2 * y * y
        return R

    @staticmethod
    def quaternion_from_rotation_matrix(rotation_matrix: np.ndarray) -> np.ndarray:
        """Shepperd-style branch on the trace / largest diagonal entry (geometry.py:216-289)."""
        _need(rotation_matrix, (3, 3), "rotation_matrix")
        _need_finite(rotation_matrix, "rotation_matrix")
        m = rotation_matrix
        if not np.allclose(m @ m.T, np.eye(3), atol=1e-6):
            raise ValidationError("rotation_matrix is not orthogonal")
        if abs(np.linalg.det(m) - 1.0) > 1e-6:
            raise ValidationError("rotation_matrix determinant is not 1")
        tr = np.trace(m)
        if tr > 0:
            s = 0.5 / np.sqrt(tr + 1.0)
            q = [0.25 / s, (m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = 2.0 * np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
            q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
        elif m[1, 1] > m[2, 2]:
            s = 2.0 * np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2])
            q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
        else:
            s = 2.0 * np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1])
            q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
        return np.array(q)

    @staticmethod
    def rotation_matrix_from_euler_angles(euler_angles: np.ndarray, convention: str = "ZYX") -> np.ndarray:
        """Rz(yaw) @ Ry(pitch) @ Rx(roll) for [roll, pitch, yaw] (geometry.py:341-390)."""
        _need(euler_angles, (3,), "euler_angles")
        _need_finite(euler_angles, "euler_angles")
        if convention != "ZYX":
            raise ValidationError(f"Only 'ZYX' convention is currently supported, got {convention}")
        roll, pitch, yaw = euler_angles
        cr, sr = np.cos(roll), np.sin(roll)
        cp, sp = np.cos(pitch), np.sin(pitch)
        cy, sy = np.cos(yaw), np.sin(yaw)
        rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        return rz @ ry @ rx

    @staticmethod
    def quaternion_from_euler_angles(euler_angles: np.ndarray, convention: str = "ZYX") -> np.ndarray:
        """Euler -> matrix -> quaternion, as the reference does (geometry.py:393-422)."""
        return GeometryUtils.quaternion_from_rotation_matrix(
            GeometryUtils.rotation_matrix_from_euler_angles(euler_angles, convention)
        )

    @staticmethod
    def euler_angles_from_rotation_matrix(rotation_matrix: np.ndarray, convention: str = "ZYX") -> np.ndarray:
        """[roll, pitch, yaw] of a ZYX matrix, with the reference's gimbal-lock branch (geometry.py:292-338)."""
        _need(rotation_matrix, (3, 3), "rotation_matrix")
        if convention != "ZYX":
            raise ValidationError(f"Only 'ZYX' convention is currently supported, got {convention}")
        m = rotation_matrix
        if abs(m[2, 0]) >= 1.0 - 1e-6:
            yaw = 0.0
            if m[2, 0] < 0:
                pitch = np.pi / 2.0
                roll = yaw + np.arctan2(m[0, 1], m[0, 2])
            else:
                pitch = -np.pi / 2.0
                roll = -yaw + np.arctan2(-m[0, 1], -m[0, 2])
        else:
            pitch = -np.arcsin(m[2, 0])
            c = np.cos(pitch)
            roll = np.arctan2(m[2, 1] / c, m[2, 2] / c)
            yaw = np.arctan2(m[1, 0] / c, m[0, 0] / c)
        return np.array([roll, pitch, yaw])

    @staticmethod
    def euler_angles_from_quaternion(quaternion: np.ndarray, convention: str = "ZYX") -> np.ndarray:
        return GeometryUtils.euler_angles_from_rotation_matrix(
            GeometryUtils.rotation_matrix_from_quaternion(quaternion), convention
        )

    @staticmethod
    def axis_angle_from_quaternion(quaternion: np.ndarray) -> Tuple[np.ndarray, float]:
        """(axis, angle) with the reference's small-angle fallbacks (geometry.py:127-165)."""
        _need(quaternion, (4,), "quaternion")
        _need_finite(quaternion, "quaternion")
        n = np.linalg.norm(quaternion)
        if n == 0:
            raise ProcessingError("Cannot convert zero quaternion to axis-angle")
        if abs(n - 1.0) > _UNIT_TOL:
            quaternion = quaternion / n
        w = quaternion[0]
        angle = 2.0 * np.arccos(np.clip(w, -1.0, 1.0))
        if abs(angle) < 1e-6:
            return np.array([1.0, 0.0, 0.0]), 0.0
        s = np.sin(angle / 2.0)
        if abs(s) < 1e-6:
            return np.array([1.0, 0.0, 0.0]), np.pi
        return GeometryUtils.normalize_vector(quaternion[1:] / s), angle

    @staticmethod
    def quaternion_angle_difference(quat_start: np.ndarray, quat_end: np.ndarray) -> float:
        """Rotation angle of q_end (x) q_start^-1 (geometry.py:456-479)."""
        delta = GeometryUtils.quaternion_multiply(quat_end, GeometryUtils.quaternion_inverse(quat_start))
        return GeometryUtils.axis_angle_from_quaternion(delta)[1]

    @staticmethod
    def validate_rotation_matrix(matrix: np.ndarray, atol: float = 1e-6) -> bool:
        if not isinstance(matrix, np.ndarray) or matrix.shape != (3, 3):
            return False
        if not np.allclose(matrix @ matrix.T, np.eye(3), atol=atol):
            return False
        return not abs(np.linalg.det(matrix) - 1.0) > atol
