"""Exception kinds of the boundary (mirrors reference src/validation_error.py:1-10)."""


class ValidationError(ValueError):
    """Bad argument: wrong shape, non-finite value, non-unit quaternion."""


class ProcessingError(RuntimeError):
    """Numerical failure while processing valid-looking input (e.g. z == 0)."""
