"""Alias of carle_amd.agents (the reference import path carle.agents)."""
from carle_amd.agents import RandomAgent, RandomNetworkAgent  # noqa: F401
