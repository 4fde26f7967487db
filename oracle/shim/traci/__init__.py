"""TEST INFRASTRUCTURE — import-time stand-in for SUMO's `traci` (never called)."""
from . import exceptions  # noqa: F401
