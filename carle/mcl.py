"""Alias of carle_amd.mcl (the reference import path carle.mcl)."""
from carle_amd.mcl import (Motivator, RND2D, AE2D, PredictionBonus, SurpriseBonus, ParsimonyBonus, MorphoBonus,  # noqa: F401
                           CornerBonus, SpeedDetector, PufferDetector, get_glider, get_morley_puffer)
