"""carle_amd -- MI355X (gfx950) native hot path of CARLE (riveSunder/carle).

The batched Life-like CA step and the RND2D / AE2D reward forward passes run in hand-written
HIP kernels inside ``carle_amd/lib/libcarle_hip.so`` (C ABI: ``include/carle_hip.h``); this
package is the Python host side that mirrors the reference's gym-style surface
(``carle_amd.env.CARLE``, ``carle_amd.mcl.{Motivator,RND2D,AE2D}``).  There is NO CPU fallback:
every compute call raises if the library or a HIP device is missing.
"""
__version__ = "0.1.0"
