"""carle_amd.rle -- RLE codec, universe logging and frame dumps.

Host-side I/O around the hot path, mirroring carle/env.py:260-513.  The per-cell Python loops of
the reference are replaced by the C++ codec in libcarle_hip.so (``carle_rle_encode_body`` /
``carle_rle_decode_body``, host pointers); output is byte-identical to the reference's
``get_rle`` (quirks included: the buffered tail of a pattern is dropped, see csrc/rle.hip).
These functions are mixed into ``CARLE`` (carle_amd/env.py).
"""
import ctypes
import os
import struct
import zlib

import numpy as np
import torch

from . import _lib


def encode_body(grid):
    """2-D array-like of cells -> RLE body string (reference line breaking)."""
    cells = np.ascontiguousarray((np.asarray(grid) != 0).astype(np.uint8))
    h, w = cells.shape
    lib = _lib.load()
    need = lib.carle_rle_encode_body(cells.ctypes.data, h, w, None, 0)
    if need < 0:
        raise ValueError("carle_rle_encode_body: bad grid")
    buf = ctypes.create_string_buffer(int(need))
    lib.carle_rle_encode_body(cells.ctypes.data, h, w, ctypes.cast(buf, ctypes.c_void_p), need)
    return buf.raw[:need].decode("ascii")


def decode_body(rle, height, width):
    """RLE body string -> (height, width) uint8 array (carle/env.py:260-328)."""
    raw = rle.encode("ascii", errors="replace")
    cells = np.zeros((height, width), np.uint8)
    rc = _lib.load().carle_rle_decode_body(raw, len(raw), cells.ctypes.data, height, width)
    if rc == 2:
        raise IndexError(_lib.load().carle_last_error().decode())
    if rc != 0:
        raise ValueError(_lib.load().carle_last_error().decode())
    return cells


def write_png_gray(path, img):
    """Minimal 8-bit grayscale PNG writer (the reference uses skimage.io.imsave, env.py:511)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class RLEMixin:
    """RLE / logging methods of the reference's CARLE (env.py:260-513)."""

    def rle_to_grid(self, rle):
        return torch.from_numpy(decode_body(rle, self.height, self.width).astype(np.float32))

    def read_rle(self, filepath):
        """Reads a .rle file: sets ``birth``/``survive`` from its ``rule = B../S..`` header and
        returns the pattern body (env.py:330-382).  Unlike the reference, a ``:T<h>, <w>`` suffix
        after the survive digits (which the reference itself writes, env.py:429) is tolerated."""
        rle, add = "", False
        with open(filepath, "r") as f:
            for line in f.readlines():
                if add:
                    rle += line
                if "rule" in line:
                    rules = line.split("/")
                    proto_birth = rules[0].split()[-1]
                    proto_survive = rules[-1].split(":")[0]
                    self.birth = sorted({int(ch) for ch in proto_birth if ch.isdigit()})
                    self.survive = sorted({int(ch) for ch in proto_survive if ch.isdigit()})
                    add = True
        return rle

    def read_csv(self, filepath):
        print("warning, read_csv not implemented yet")
        return ""

    def load_universe(self, filepath, universe_index=0):
        my_rle = self.read_rle(filepath) if "rle" in filepath[-4:] else self.read_csv(filepath)
        my_universe = self.rle_to_grid(my_rle)
        u = self.universe
        assert u.shape[2] == my_universe.shape[0] and u.shape[3] == my_universe.shape[1], \
            "tried to load the wrong size universe"
        u = u.clone()
        u[universe_index, 0, :, :] = my_universe.to(u.device)
        self.universe = u

    def get_rle(self, universe, action=False):
        if isinstance(universe, torch.Tensor):
            universe = universe.detach().squeeze().cpu().numpy()
        else:
            universe = np.asarray(universe).squeeze()
        rle = "#C exp_id={} \n".format(self.instance_id)
        rle += "#C step={} ({}) \n".format(self.step_number, "action" if action else "universe")
        rle += "x = 0, y = 0, rule = B" + "".join(str(b) for b in self.birth)
        rle += "/S" + "".join(str(s) for s in self.survive)
        rle += ":T{}, {}\n".format(self.height, self.width)
        return rle + encode_body(universe)

    def log_universe(self, universe_index=0):
        rle_universe = self.get_rle(self.universe[universe_index, 0, :, :])
        action = self.action
        if getattr(action, "materialize", None) is not None:
            action = action.materialize()
        action = torch.as_tensor(action)
        while action.dim() < 4:
            action = action.unsqueeze(0)
        rle_action = self.get_rle(action[min(universe_index, action.shape[0] - 1), 0, :, :], action=True)
        self.log.append([rle_action, rle_universe])

    def save_log(self, directory="./logs"):
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "carle_log{}.csv".format(self.instance_id)), "w") as f:
            f.write("action,universe,\n")
            for log_entry in self.log:
                for entry in log_entry:
                    f.write('"' + entry + '"' + ",")
                f.write("\n")

    def save_rle(self, rle, directory="./logs"):
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "universe{}_step{}.rle".format(self.instance_id, self.step_number)), "w") as f:
            f.write(rle)

    def save_frame(self, directory="./frames"):
        os.makedirs(directory, exist_ok=True)
        img = np.uint8(255 * self.universe[0, 0, :, :].detach().cpu().numpy())
        write_png_gray(os.path.join(directory, "frame{}_step{}.png".format(self.instance_id, self.step_number)), img)
