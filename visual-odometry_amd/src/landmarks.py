"""Landmark generator: only ``generate_random`` is restated (reference src/landmarks.py:9-90),
because the BA test scene, the demo scene and the golden fixtures are built from it."""

from typing import Optional, Tuple

import numpy as np

from .validation_error import ProcessingError, ValidationError


class LandmarkGenerator:
    def __init__(self, bounds_x: Tuple[float, float] = (-10.0, 10.0),
                 bounds_y: Tuple[float, float] = (-10.0, 10.0),
                 bounds_z: Tuple[float, float] = (0.0, 5.0)) -> None:
        for name, b in (("bounds_x", bounds_x), ("bounds_y", bounds_y), ("bounds_z", bounds_z)):
            if not isinstance(b, tuple) or len(b) != 2:
                raise ValidationError(f"{name} must be a tuple of (min, max), got {b}")
            for label, v in zip(("min", "max"), b):
                if not isinstance(v, (int, float)) or not np.isfinite(v):
                    raise ValidationError(f"{name} {label} value must be a finite number, got {v}")
            if b[0] >= b[1]:
                raise ValidationError(f"{name} min ({b[0]}) must be less than max ({b[1]})")
        self.bounds_x, self.bounds_y, self.bounds_z = bounds_x, bounds_y, bounds_z

    def generate_random(self, num_landmarks: int, seed: Optional[int] = None) -> np.ndarray:
        """(N, 3) uniform landmarks.  Draw order matters for reproducibility: all x, then all y,
        then all z from one ``default_rng(seed)`` (reference landmarks.py:76-84)."""
        if not isinstance(num_landmarks, int) or num_landmarks <= 0:
            raise ValidationError(f"num_landmarks must be a positive integer, got {num_landmarks}")
        if num_landmarks > 1_000_000:
            raise ValidationError(f"num_landmarks too large ({num_landmarks}), maximum allowed is 1,000,000")
        rng = np.random.default_rng(seed)
        cols = [rng.uniform(lo, hi, num_landmarks) for lo, hi in (self.bounds_x, self.bounds_y, self.bounds_z)]
        pts = np.column_stack(cols)
        if not np.all(np.isfinite(pts)):
            raise ProcessingError("Generated landmarks contain non-finite values")
        return pts
