"""TEST INFRASTRUCTURE — `from tensorflow.keras.layers import Dense` (see keras/__init__.py)."""
from . import Dense, Layer  # noqa: F401
