"""oracle/prediction.py -- CPU checker for PredictionBonus / SurpriseBonus.  TEST INFRASTRUCTURE.

Restatement (torch CPU, fp32, autograd + torch.optim.Adam, eval mode: dropout = identity) of the two AE2D subclasses of
the reference (riveSunder/carle):

  PredictionBonus  carle/mcl.py:558-699  the AE2D predictor applied to the OLDEST frame of a buffer of at most
                                         prediction_steps + 1 = 6 frames, squared error against the newest;
                                         reward += scale * (0.1 - error), zero for grids without a live cell;
                                         update every batch_size steps on the mean loss of the window (Adam lr 6e-2: the
                                         class's 3e-4 is assigned after the optimiser was built, mcl.py:563), gradients
                                         zeroed at the START of a window only (AE2D.update_predictor, mcl.py:498-502)
  SurpriseBonus    carle/mcl.py:701-728  AE2D's reconstruction error as the reward, same zeroing and update rule

The environment is not part of it: ``step`` takes the observation the CA oracle produced (oracle/ca.py).

Parity status: PINNED against tests/golden/prediction.npz, recorded from the genuine reference by
tools/make_golden.py::make_prediction (tests/test_oracle_nets.py).
"""
import numpy as np
import torch

from . import nets

_AE_KEYS = ("0.weight", "0.bias", "4.weight", "4.bias", "8.weight", "8.bias", "11.weight", "11.bias")


class _AEWrapperOracle:
    def __init__(self, sd, batch_size=64, reward_scale=1.0, prefix="predictor."):
        self.sd = {"predictor." + k: torch.tensor(np.asarray(sd[prefix + k]), dtype=torch.float32, requires_grad=True)
                   for k in _AE_KEYS}
        self.optimizer = torch.optim.Adam([self.sd["predictor." + k] for k in _AE_KEYS], lr=6e-2)   # mcl.py:243,480-481
        self.batch_size = batch_size
        self.reward_scale = reward_scale
        self.buffer_length = 0
        self.accumulate_loss = 0.0
        self.live_cells = None

    def state(self):
        return {k: v.detach().numpy().copy() for k, v in self.sd.items()}

    def _zero_grad(self):
        for p in self.sd.values():
            p.grad = None

    def _update(self, loss):                                  # AE2D.update_predictor, mcl.py:498-502: no zero_grad after
        loss.backward()
        self.optimizer.step()

    def loss(self, obs):
        raise NotImplementedError

    def get_bonus(self, obs):
        return self.loss(obs)

    def get_bonus_update(self, obs):                          # mcl.py:644-656
        self._zero_grad()
        loss = self.loss(obs)
        self._update(torch.mean(loss))
        return loss.detach()

    def get_bonus_accumulate(self, obs):                      # mcl.py:658-677 / 528-547
        if self.buffer_length == 0:
            self._zero_grad()
            self.accumulate_loss = 0.0
        loss = self.loss(obs)
        self.accumulate_loss = self.accumulate_loss + torch.mean(loss)
        self.buffer_length += 1
        if self.buffer_length >= self.batch_size:
            self._update(self.accumulate_loss / self.batch_size)
            self.buffer_length = 0
        return loss.detach()

    def _bonus_of(self, err):
        raise NotImplementedError

    def step(self, obs, reward):
        """obs: (N,1,H,W) float32 grids after the environment step; reward: (N,1) float32 -> new reward."""
        o = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
        err = self.get_bonus_accumulate(o).unsqueeze(1)
        bonus = self._bonus_of(err).clone()
        empty = o.mean(dim=[1, 2, 3]) == 0.0                  # mcl.py:687-691
        bonus[empty] = bonus[empty] * 0.0
        self.live_cells = float(o.sum())
        return (torch.from_numpy(np.asarray(reward, dtype=np.float32)) + np.float32(self.reward_scale) * bonus).numpy()


class PredictionBonus(_AEWrapperOracle):
    prediction_steps = 5

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.grid_buffer = []

    def loss(self, obs):                                      # mcl.py:581-605
        o = obs if isinstance(obs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
        self.grid_buffer.append(o)
        prediction = nets.ae_predictor(self.grid_buffer[0], self.sd)
        target = self.grid_buffer[-1]
        loss = torch.mean(torch.abs((target - prediction) ** 2), dim=[1, 2, 3])
        if len(self.grid_buffer) > self.prediction_steps:
            self.grid_buffer.pop(0)
        return loss

    def _bonus_of(self, err):
        return 0.1 - err


class SurpriseBonus(_AEWrapperOracle):
    def loss(self, obs):                                      # AE2D.get_bonus, mcl.py:504-512
        o = obs if isinstance(obs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32))
        return torch.mean(torch.abs((o - nets.ae_predictor(o, self.sd)) ** 2), dim=[1, 2, 3])

    def _bonus_of(self, err):
        return err
