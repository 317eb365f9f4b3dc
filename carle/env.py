"""Alias of carle_amd.env (the reference import path carle.env)."""
from carle_amd.env import CARLE, CARLEPacked  # noqa: F401
