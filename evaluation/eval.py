"""Alias of carle_amd.harness.evaluate (the reference import path evaluation.eval)."""
from carle_amd.harness import evaluate  # noqa: F401
