"""Import-path alias: ``carle`` -> carle_amd, so code written against riveSunder/carle
(``from carle.env import CARLE``, ``from carle.mcl import RND2D, AE2D``) runs on the HIP path."""
