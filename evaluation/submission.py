"""Alias of carle_amd.agents (the reference import path evaluation.submission)."""
from carle_amd.agents import DemoAgent, SubmissionAgent  # noqa: F401
