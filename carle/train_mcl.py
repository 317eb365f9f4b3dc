"""Alias of carle_amd.harness.train (the reference import path carle.train_mcl)."""
from carle_amd.harness import train  # noqa: F401
