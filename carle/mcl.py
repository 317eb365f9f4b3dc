"""Alias of carle_amd.mcl (the reference import path carle.mcl)."""
from carle_amd.mcl import (Motivator, RND2D, AE2D, ParsimonyBonus, CornerBonus, SpeedDetector,  # noqa: F401
                           PufferDetector, get_glider, get_morley_puffer)
