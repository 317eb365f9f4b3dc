"""oracle/torch_port.py -- PyTorch-CPU port of CARLE.step.  TEST INFRASTRUCTURE, NOT PRODUCT.

Replays the ATen op sequence the reference executes per step (carle/env.py:188-242: two
reductions over the action, zero-pad, logical_xor, circular 3x3 conv, one ``eq`` per rule digit,
blend) so that bench.py's ``cpu_baseline`` leg (kind "port") times the same work the reference
does on the host cores.  The reference itself cannot travel to the GPU box.

Parity status: PINNED -- tests/test_oracle.py::test_torch_port_matches_golden replays golden
cases through this port bit for bit.
"""
from functools import reduce

import torch
import torch.nn.functional as F


class TorchPortEnv:
    """Minimal stateful port: universe (N,1,H,W) fp32 on CPU, reference op order per step."""

    def __init__(self, n, size, action_size, birth=(3,), survive=(2, 3)):
        self.n, self.size = n, size
        self.birth, self.survive = list(birth), list(survive)
        self.kernel = torch.tensor([[1., 1., 1.], [1., 0., 1.], [1., 1., 1.]]).reshape(1, 1, 3, 3)
        p = (size - action_size) // 2
        self.pad = (p, p, p, p)                       # env.py:127-132
        self.universe = torch.zeros(n, 1, size, size)
        self.step_number = 0

    def step(self, action):
        any_toggle = bool(torch.sum(action))          # env.py:191 (host sync in the reference)
        del any_toggle
        padded = F.pad(action, self.pad)              # env.py:179
        self.universe = 1.0 * torch.logical_xor(self.universe, padded)       # env.py:182
        if torch.mean(action) == 1.0:                 # env.py:208
            self.universe = torch.zeros(self.n, 1, self.size, self.size)
            self.step_number = 0
        else:
            nb = F.conv2d(F.pad(self.universe, (1, 1, 1, 1), mode="circular"), self.kernel)   # env.py:219
            born = reduce(lambda a, b: 1.0 * (a + b), [k == nb for k in self.birth])       # env.py:221
            stay = reduce(lambda a, b: 1.0 * (a + b), [k == nb for k in self.survive])     # env.py:223
            self.universe = (1 - self.universe) * born + self.universe * stay             # env.py:226
            self.step_number += 1
        return self.universe, torch.zeros(self.n, 1), torch.zeros(self.n, 1), [{}] * self.n
