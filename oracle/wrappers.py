"""oracle/wrappers.py -- CPU checker for the hand-coded reward wrappers.  TEST INFRASTRUCTURE.

numpy restatement (fp32 arithmetic where the reference computes in fp32) of the reward wrappers of the reference
(riveSunder/carle, carle/mcl.py) that evaluation/eval.py:83-87 and carle/train_mcl.py stack around the environment:

  ParsimonyBonus   mcl.py:86-105    reward = 100 * reward / max(sum(action) per instance, 100)
  CornerBonus      mcl.py:197-231   reward += scale * sum(reward_mask * obs) + scale * sum(punish_mask * obs)
  SpeedDetector    mcl.py:730-799   centre of mass outside the action window, velocity, ONE batch-coupled speed
                                    (sqrt of the sum over all instances and both axes), reward += speed
  PufferDetector   mcl.py:804-853   list of live-cell totals over action-free steps, reward += 1 when the total grew over
                                    more than `growth_threshold` such steps; any action clears the list

Each class is stepped with what its wrapped environment returned: ``step(obs, reward, action)`` -> reward (the wrappers
leave ``obs`` alone).  ``obs`` is the inner environment's universe after the step (the reference reads
``inner_env.universe``, which is the observation object, env.py:229-236).

Parity status: PINNED -- against the reference-recorded rewards of tests/golden/wrappers.npz
(tests/test_oracle.py::test_wrappers_oracle_reproduces_the_reference_fixtures) and, in the build container, against the
genuine reference on open-ended random cases (tests/tools/fuzz_oracle_vs_reference.py).
"""
import numpy as np

F = np.float32


def _f32(x):
    return np.asarray(x, dtype=np.float32)


class ParsimonyBonus:
    """mcl.py:86-105.  ``action.sum(axis=[1,2,3])`` is an fp32 sum per instance; the maximum with the one-element
    tensor [100.0] broadcasts, so the result has shape (N,) and the division broadcasts reward (N,1) against (N,)
    -> (N,N) for N > 1 in the reference (a quirk; the build keeps the per-instance meaning):
    this restatement follows the reference for N = 1 and the per-instance meaning beyond."""

    def step(self, obs, reward, action):
        a = _f32(action)
        s = a.reshape(a.shape[0], -1).sum(axis=1, dtype=np.float32)
        return (F(100.0) * _f32(reward) / np.maximum(s, F(100.0)).reshape(-1, 1)).astype(np.float32)


class CornerBonus:
    """mcl.py:197-231 (masks :206-218: the python slices ii-4:ii+4 with negative starts select nothing for ii < 4)."""

    def __init__(self, height, width, reward_scale=1.0):
        self.reward_scale = F(reward_scale)
        self.reward_mask = np.zeros((1, 1, height, width), np.float32)
        self.punish_mask = np.zeros((1, 1, height, width), np.float32)
        self.reward_mask[:, :, :16, :16] = 1.0
        for ii in range(96):
            self.reward_mask[:, :, ii - 4:ii + 4, ii - 4:ii + 4] = 1.0
        self.punish_mask[:, :, -64:, -64:] = -1.0
        self.punish_mask[:, :, :64, -64:] = -1.0

    def step(self, obs, reward, action):
        o = _f32(obs)
        r = _f32(reward).copy()
        # (mask * obs).sum(axis=-1).sum(axis=-1) -> (N,1); sums of 0 / 1 values: exact in fp32
        r += self.reward_scale * (self.reward_mask * o).sum(axis=-1, dtype=np.float32).sum(axis=-1, dtype=np.float32)
        r += self.reward_scale * (self.punish_mask * o).sum(axis=-1, dtype=np.float32).sum(axis=-1, dtype=np.float32)
        return r


class SpeedDetector:
    """mcl.py:730-799.  Weights :742-757: mass_weight_w = arange(height) as a ROW (it varies along the last axis),
    mass_weight_h = arange(width) as a COLUMN, both zeroed inside the centred action window.  The reward update happens
    from the second step on (:781-793); ``speed`` is ONE scalar for the whole batch (:789)."""

    def __init__(self, height, width, action_height, action_width):
        mask = np.ones((1, 1, height, width), np.float32)
        oy, ox = (height - action_height) // 2, (width - action_width) // 2
        mask[:, :, oy:oy + action_height, ox:ox + action_width] = 0.0
        self.mass_weight_w = np.arange(height, dtype=np.float32).reshape(1, -1) * mask
        self.mass_weight_h = np.arange(width, dtype=np.float32).reshape(-1, 1) * mask
        self.center_of_mass = None
        self.speed = None
        self.velocity = None
        self.live_cells = None

    def step(self, obs, reward, action):
        o = _f32(obs)
        n = o.shape[0]
        r = _f32(reward).copy()
        # sums of integers: exact in fp32 in any order while they stay below 2^24 (live cells: N x H x W up to 4096 x 64 x 64;
        # weighted sums: up to 256 x 256 grids); accumulated in fp64 here and rounded once -- beyond that the reference's
        # own fp32 summation order decides the last bits, and the comparison tolerances (1e-4) allow for it
        live = o.reshape(n, -1).sum(axis=1, dtype=np.float64).astype(np.float32)
        s_h = (o * self.mass_weight_h).reshape(n, -1).sum(axis=1, dtype=np.float64).astype(np.float32)
        s_w = (o * self.mass_weight_w).reshape(n, -1).sum(axis=1, dtype=np.float64).astype(np.float32)
        com = np.stack([s_h / (live + F(1e-7)), s_w / (live + F(1e-7))]).astype(np.float32)      # fp32, as the reference
        if self.center_of_mass is None:
            self.center_of_mass = com
        else:
            velocity = (self.center_of_mass - com).astype(np.float32)
            speed = np.sqrt(np.sum(velocity * velocity, dtype=np.float32), dtype=np.float32)
            self.speed, self.velocity, self.center_of_mass = speed, velocity, com
            r = (r + speed).astype(np.float32)
        self.live_cells = live
        return r


class PufferDetector:
    """mcl.py:804-853.  ``not(torch.sum(action))``: the fp32 sum of the action tensor decides (values that cancel count
    as "no action"); ``live_cells`` is the total over the WHOLE batch (:832)."""

    def __init__(self, growth_threshold=512):
        self.growth_threshold = growth_threshold
        self.cells = []
        self.live_cells = 0.0

    def step(self, obs, reward, action, action_free=None):
        """``action_free``: override of the predicate (the build's default is "no element non-zero", identical for
        0 / 1 actions: DESIGN.md section 10)."""
        r = _f32(reward).copy()
        self.live_cells = float(_f32(obs).sum(dtype=np.float64))
        free = (not bool(_f32(action).sum(dtype=np.float32))) if action_free is None else action_free
        if free:
            self.cells.append(self.live_cells)
            if len(self.cells) > self.growth_threshold:
                slope = self.cells[-1] - self.cells[0]
                self.cells.pop(0)
                if slope > 0.01:
                    r = (r + F(1.0)).astype(np.float32)
        else:
            self.cells = []
        return r


class MorphoBonus:
    """mcl.py:107-195.  ``patterns``: the (K,1,8,8) target kernels (what ``add_rle_pattern`` builds, mcl.py:144-167:
    ``make_patterns`` below restates it from a decoded RLE grid).  ``step`` takes the grid BEFORE the environment step and
    the action (same shape), and the reward the wrapped environment returned: reward += scale * (max + min) of
    conv2d(|grid - action|, patterns) over all patterns and positions, per instance (mcl.py:172-181).  The correlation is a
    plain fp64 sum here (the reference: ATen's fp32 conv2d; the comparison carries a 1e-5 tolerance)."""

    def __init__(self, patterns, reward_scale=1.0):
        self.patterns = np.asarray(patterns, dtype=np.float64).reshape(-1, 8, 8)
        self.reward_scale = F(reward_scale)

    @staticmethod
    def make_patterns(grid, dim=8):
        """grid: the (H,W) 0/1 array ``rle_to_grid`` returns -> the six oriented kernels (6,1,dim,dim), float32."""
        g = np.pad(_f32(grid), ((2, 1), (1, 1)))[:dim, :dim].copy()          # ZeroPad2d((1,1,2,1)): left, right, top, bottom
        g[g == 0] = -1
        live = g == 1
        # ``15. / tensor`` is torch's __rtruediv__: reciprocal first, then the product (one ulp from 15 / sum at 14 cells:
        # found by tests/tools/fuzz_oracle_vs_reference.py)
        g[live] = g[live] * F(F(F(1.0) / g[live].sum(dtype=np.float32)) * F(15.0))
        p = g[None, None]
        t = np.transpose(p, (0, 1, 3, 2))
        return np.concatenate([p, p[:, :, ::-1], p[:, :, :, ::-1], t[:, :, ::-1], t[:, :, :, ::-1], t]).astype(np.float32)

    def match(self, grid_before, action):
        g = np.abs(_f32(grid_before) - _f32(action)).astype(np.float64)[:, 0]
        n, h, w = g.shape
        win = np.lib.stride_tricks.sliding_window_view(g, (8, 8), axis=(1, 2))        # (n, h-7, w-7, 8, 8)
        conv = np.einsum("nyxij,kij->nkyx", win, self.patterns)
        return conv.reshape(n, -1).max(axis=1), conv.reshape(n, -1).min(axis=1)

    def step(self, grid_before, reward, action):
        mx, mn = self.match(grid_before, action)
        return _f32(reward) + self.reward_scale * (_f32(mx) + _f32(mn))[:, None]
