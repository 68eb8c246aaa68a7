"""Camera pose value type (position + unit quaternion [w, x, y, z] + timestamp).

Own restatement of reference ``src/camera_pose.py:10-260`` (``CameraPose``).  The world -> camera
transform is the row form ``(P - t) @ R(q)`` (camera_pose.py:85-86), i.e. Xc = R^T (P - t) with
R rotating camera -> world; this is the transform the HIP kernels restate.
"""

from typing import Optional

import numpy as np

from .geometry import GeometryUtils
from .validation_error import ValidationError
from .validation_helper import ValidationHelper


class CameraPose:
    def __init__(self, position: np.ndarray, orientation: np.ndarray, timestamp: float = 0.0):
        ValidationHelper.validate_pt3d(position)
        if not GeometryUtils.validate_quaternion(orientation):
            raise ValidationError("orientation quaternion is not valid")
        self.position = position.copy()
        self.orientation_quaternion = orientation.copy()
        self.timestamp = timestamp

    @property
    def rotation_matrix(self) -> np.ndarray:
        return GeometryUtils.rotation_matrix_from_quaternion(self.orientation_quaternion)

    @property
    def quaternion(self) -> np.ndarray:
        return self.orientation_quaternion.copy()

    def transform_points_world_to_camera(self, points_world: np.ndarray) -> np.ndarray:
        ValidationHelper.validate_pts3d(points_world)
        return (points_world - self.position.reshape(1, 3)) @ self.rotation_matrix

    def transform_points_camera_to_world(self, points_camera: np.ndarray) -> np.ndarray:
        ValidationHelper.validate_pts3d(points_camera)
        return (self.rotation_matrix @ points_camera.T).T + self.position

    def convert_to_new_frame(self, camera_pose: "CameraPose") -> "CameraPose":
        """Express this pose in the frame of ``camera_pose`` (camera_pose.py:35-55)."""
        p = camera_pose.transform_points_world_to_camera(self.position.reshape(1, 3)).reshape(3)
        q = GeometryUtils.quaternion_multiply(
            GeometryUtils.quaternion_inverse(camera_pose.quaternion), self.quaternion
        )
        return CameraPose(p, q, self.timestamp)

    def get_transform_matrix(self) -> np.ndarray:
        """4x4 world -> camera matrix [R^T, -R^T t; 0, 1] (camera_pose.py:104-116)."""
        rt = self.rotation_matrix.T
        m = np.eye(4)
        m[:3, :3] = rt
        m[:3, 3] = -rt @ self.position
        return m

    @classmethod
    def from_transform_matrix(cls, transform_matrix: np.ndarray,
                              timestamp: Optional[float] = None) -> "CameraPose":
        if not isinstance(transform_matrix, np.ndarray) or transform_matrix.shape != (4, 4):
            got = transform_matrix.shape if isinstance(transform_matrix, np.ndarray) else type(transform_matrix)
            raise ValidationError(f"transform_matrix must be a 4x4 numpy array, got {got}")
        r = transform_matrix[:3, :3]
        return cls(-r.T @ transform_matrix[:3, 3], GeometryUtils.quaternion_from_rotation_matrix(r), timestamp)

    @staticmethod
    def create_look_straight_pose(initial_position: Optional[np.ndarray] = None,
                                  timestamp: float = 0.0) -> "CameraPose":
        """Camera z-axis along world x (camera_pose.py:218-242); the timestamp is always 0.0 there."""
        if initial_position is None:
            initial_position = np.array([0.0, 0.0, 0.0])
        q = GeometryUtils.quaternion_from_euler_angles(np.array([-np.pi / 2, 0.0, -np.pi / 2]))
        return CameraPose(position=initial_position, orientation=q, timestamp=0.0)

    def copy(self) -> "CameraPose":
        return CameraPose(self.position.copy(), self.orientation_quaternion.copy(), self.timestamp)

    def __repr__(self) -> str:
        t = f", t={self.timestamp:.3f}" if self.timestamp is not None else ""
        p, q = self.position, self.orientation_quaternion
        return (f"CameraPose(position=[{p[0]:.3f}, {p[1]:.3f}, {p[2]:.3f}], "
                f"orientation=[{q[0]:.3f}, {q[1]:.3f}, {q[2]:.3f}, {q[3]:.3f}]{t})")
