"""Argument / return containers of the BA entry point: Features2D, Landmarks3D, ImageFeatures.

Own restatement of reference ``src/feature_observation.py:13-289``.  Getters return copies, as in
the reference (:34,42,98,106).  The drawing helper ``to_opencv_image`` (the only cv2 user) and the
FeatureObservation/FeatureTrack/Tracks/ImageObservations family (:292-544) are outside the hot
path and are not provided.
"""

from typing import Optional

import numpy as np

from .camera import PinHoleCamera
from .validation_helper import ValidationHelper


class _IdTable:
    """N rows of coordinates + N integer ids; base for Features2D / Landmarks3D."""

    _attr = "points"

    def _rows(self) -> np.ndarray:
        return getattr(self, self._attr)

    def get_ids(self) -> np.ndarray:
        return self.ids.copy()

    def _select(self, ids: np.ndarray):
        # np.intersect1d returns the common ids SORTED; rows follow that order
        # (reference feature_observation.py:53-56,117-120 -- SURVEY.md Appendix B.1).
        common, mine, _ = np.intersect1d(self.ids, ids, return_indices=True)
        return self._rows()[mine], common

    def add_noise(self, noise_std: float) -> None:
        if noise_std <= 0.0:
            return
        rows = self._rows()
        rows += np.random.randn(len(rows), rows.shape[1]) * noise_std


class Features2D(_IdTable):
    _attr = "points_2d"

    def __init__(self, points_2d: np.ndarray, ids: np.ndarray):
        ValidationHelper.validate_pts2d(points_2d)
        ValidationHelper.validate_ids(ids)
        self.points_2d = points_2d.copy()
        self.ids = ids.copy()

    def get_points_2d(self) -> np.ndarray:
        return self.points_2d.copy()

    def get_selected_ids(self, ids: np.ndarray) -> "Features2D":
        return Features2D(*self._select(ids))

    def copy(self) -> "Features2D":
        return Features2D(self.points_2d, self.ids)


class Landmarks3D(_IdTable):
    _attr = "points_3d"

    def __init__(self, points_3d: np.ndarray, ids: np.ndarray):
        ValidationHelper.validate_pts3d(points_3d)
        ValidationHelper.validate_ids(ids)
        self.points_3d = points_3d.copy()
        self.ids = ids.copy()

    def get_points_3d(self) -> np.ndarray:
        return self.points_3d.copy()

    def get_selected_ids(self, ids: np.ndarray) -> "Landmarks3D":
        return Landmarks3D(*self._select(ids))

    def copy(self) -> "Landmarks3D":
        return Landmarks3D(self.points_3d, self.ids)


class ImageFeatures:
    """2-D observations of one image: pass-through metadata + a Features2D."""

    def __init__(self, timestamp: float, camera_id: int, image_width: int, image_height: int,
                 features_2d: Optional[Features2D] = None):
        self.timestamp = timestamp
        self.camera_id = camera_id
        self.image_width = image_width
        self.image_height = image_height
        if features_2d is None:
            features_2d = Features2D(np.empty((0, 2)), np.empty((0,), dtype=int))
        self.features_2d = features_2d

    def get_timestamp(self) -> float:
        return self.timestamp

    def get_camera_id(self) -> int:
        return self.camera_id

    def get_points_2d(self) -> Features2D:
        return self.features_2d

    @classmethod
    def from_points_3d(cls, timestamp: float, camera_id: int, image_width: int, image_height: int,
                       camera: PinHoleCamera, points_3d: np.ndarray, ids: np.ndarray) -> "ImageFeatures":
        """Project camera-frame points and wrap them (reference feature_observation.py:256-289)."""
        ValidationHelper.validate_pts3d(points_3d)
        ValidationHelper.validate_ids(ids)
        return cls(timestamp, camera_id, image_width, image_height,
                   Features2D(camera.project(points_3d), ids))
