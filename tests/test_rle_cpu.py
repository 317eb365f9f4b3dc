"""CPU tests of the C++ RLE codec against strings / grids recorded from the genuine reference
(tests/golden/rle.json): byte-identical get_rle output, identical rle_to_grid grids."""
import base64
import json
import os

import numpy as np
import pytest
import torch

import golden_util as G


def _bits(b64, size):
    raw = np.frombuffer(base64.b64decode(b64), dtype=np.uint8)
    return np.unpackbits(raw, count=size * size, bitorder="little").reshape(size, size)


def _cases():
    return json.load(open(os.path.join(G.GOLDEN, "rle.json")))


def test_get_rle_is_byte_identical_to_the_reference():
    from carle_amd.env import CARLE
    for c in _cases()["encode"]:
        env = CARLE(width=c["size"], height=c["size"], device="cpu")
        env.birth, env.survive = c["birth"], c["survive"]
        env.instance_id = c["instance_id"]
        env.__dict__["_fake_step"] = c["step"]
        type(env).step_number.fget                      # property exists
        grid = torch.from_numpy(_bits(c["grid"], c["size"]).astype(np.float32))
        # step_number lives on the device in this build; patch the read for the header
        env_cls = type("E", (type(env),), {"step_number": property(lambda self: self.__dict__["_fake_step"])})
        env.__class__ = env_cls
        assert env.get_rle(grid, action=c["action"]) == c["rle"]
        body = c["rle"][c["rle"].index("\n", c["rle"].index("rule")) + 1:]
        back = env.rle_to_grid(body).numpy()
        assert np.array_equal(back != 0, _bits(c["decoded"], c["size"]) != 0)


def test_rle_to_grid_matches_the_reference():
    from carle_amd.env import CARLE
    for c in _cases()["decode"]:
        env = CARLE(width=c["size"], height=c["size"], device="cpu")
        got = env.rle_to_grid(c["rle"]).numpy()
        assert np.array_equal(got != 0, _bits(c["grid"], c["size"]) != 0), c["rle"]
    env = CARLE(width=8, height=8, device="cpu")
    with pytest.raises(IndexError):
        env.rle_to_grid("o$o$o$o$o$o$o$o$o$o!")       # taller than the grid, as in the reference
    with pytest.raises(ValueError):
        env.rle_to_grid("x3o!")


def test_read_rle_and_png(tmp_path):
    from carle_amd.env import CARLE
    from carle_amd.rle import write_png_gray
    p = tmp_path / "glider.rle"
    p.write_text("#C a glider\nx = 3, y = 3, rule = B3/S23\nbo$2bo$3o!\n")
    env = CARLE(width=64, height=64, device="cpu")
    body = env.read_rle(str(p))
    assert env.birth == [3] and env.survive == [2, 3]
    g = env.rle_to_grid(body).numpy()
    assert sorted(map(tuple, np.argwhere(g != 0).tolist())) == [(0, 1), (1, 2), (2, 0), (2, 1), (2, 2)]
    q = tmp_path / "own.rle"
    q.write_text("x = 0, y = 0, rule = B368/S245:T64, 64\n3o!\n")
    env.read_rle(str(q))
    assert env.birth == [3, 6, 8] and env.survive == [2, 4, 5]
    write_png_gray(str(tmp_path / "f.png"), np.uint8(255 * (np.arange(64).reshape(8, 8) % 2)))
    assert (tmp_path / "f.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
