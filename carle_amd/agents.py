"""carle_amd.agents -- device-side agents.

Mirrors ``carle.agents.{RandomAgent, RandomNetworkAgent}`` of the reference
(carle/agents.py:15-102) and ``evaluation.submission.{DemoAgent, SubmissionAgent}``
(evaluation/submission.py:14-50): same kwargs, ``agent(obs) -> action (N,1,AW,AH)`` of 0/1 floats.

The reference draws the toggles with ``torch.rand`` on the CPU and the wrappers copy the action to
the GPU every step (64 MiB over PCIe at N=4096).  Here:
  * ``RandomAgent`` draws on the device with the HIP Philox kernel ``carle_bernoulli_f32``;
  * ``RandomAgent(fused=True)`` returns a ``DeviceRandomAction`` token instead of a tensor: the
    environment then draws the same toggles INSIDE the CA step kernel and never reads (or writes)
    an action tensor at all.  ``token.materialize()`` yields exactly the tensor the kernel used.
The reference's agent is unseeded, so parity with it is distributional (toggle rate); bit-level
parity is against oracle/philox.py.
"""
import torch
import torch.nn as nn

from . import hip


class DeviceRandomAction:
    """A Bernoulli(p) action that has not been materialised: (p, seed, offset) identify it."""
    __slots__ = ("p", "seed", "offset", "shape", "device")

    def __init__(self, p, seed, offset, shape, device):
        self.p, self.seed, self.offset, self.shape, self.device = p, seed, offset, tuple(shape), device

    def materialize(self):
        return hip.bernoulli_f32(self.shape, self.p, self.seed, self.offset, self.device)


class RandomAgent(nn.Module):
    """Bernoulli(0.1) toggles (carle/agents.py:15-42), drawn on the observation's device."""

    def __init__(self, **kwargs):
        super().__init__()
        self.action_width = kwargs.get("action_width", 64)
        self.action_height = kwargs.get("action_height", 64)
        self.observation_width = kwargs.get("observation_width", 256)
        self.observation_height = kwargs.get("observation_height", 256)
        self.toggle_rate = 0.100
        self.seed = int(kwargs.get("seed", 0))
        self.fused = bool(kwargs.get("fused", False))
        self.calls = 0                       # Philox stream offset: one per forward call

    def forward(self, obs):
        instances = obs.shape[0]
        shape = (instances, 1, self.action_width, self.action_height)
        device = obs.device if obs.is_cuda else torch.device("cuda", torch.cuda.current_device())
        token = DeviceRandomAction(self.toggle_rate, self.seed, self.calls, shape, device)
        self.calls += 1
        return token if self.fused else token.materialize()


class RandomNetworkAgent(nn.Module):
    """Frozen random convnet -> sigmoid <= toggle_rate (carle/agents.py:45-102).  ``network`` is the
    reference's ``nn.Sequential`` (same indices: its state_dict loads strictly; it is never evaluated); the forward pass runs
    in two HIP kernels (``carle_agent_rn_forward_f32``: tiled convolutions + pooling, then the linear
    layer on the fp32 matrix cores with sigmoid and threshold fused) on the observation's device.
    Deviation: the reference reads ``observation_width`` only when the misspelt key
    ``observatoin_width`` is present (agents.py:56-58) and otherwise builds for width 256; here
    either spelling sets it."""

    def __init__(self, **kwargs):
        super().__init__()
        self.action_width = kwargs.get("action_width", 64)
        self.action_height = kwargs.get("action_height", 64)
        self.observation_width = kwargs.get("observation_width", kwargs.get("observatoin_width", 256))
        self.observation_height = kwargs.get("observation_height", 256)
        self.depth = 3
        self.filter_dim = 4
        self.toggle_rate = 0.1
        self.initialize_network()

    def initialize_network(self):
        dense_nodes = (self.observation_width // 4) * (self.observation_height // 4)
        output_nodes = self.action_width * self.action_height
        self.network = nn.Sequential(
            nn.Conv2d(1, self.filter_dim, 3, padding=1, bias=False),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Conv2d(self.filter_dim, 1, 3, padding=1, bias=False),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Flatten(),
            nn.Linear(dense_nodes, output_nodes, bias=False),
            nn.Sigmoid())
        for param in self.network.parameters():
            param.requires_grad = False

    def _kernel_domain(self, obs):
        """What carle_agent_rn_forward_f32 covers: H, W multiples of 8, fp32 weights, at most 65535 instances."""
        h, w = obs.shape[2], obs.shape[3]
        return (h % 8 == 0 and w % 8 == 0 and obs.shape[0] <= 65535 and
                all(self.network[i].weight.dtype == torch.float32 for i in (0, 3, 7)))

    def forward(self, obs, want_output=False):
        hip.require_gpu()                    # no CPU path: a missing device / library raises, it is never worked around
        if not torch.is_tensor(obs):
            obs = torch.as_tensor(obs)
        if not obs.is_cuda:
            obs = obs.to(torch.device("cuda", torch.cuda.current_device()))
        obs = obs.to(torch.float32).contiguous()
        if obs.dim() == 3:
            obs = obs.unsqueeze(1)
        instances = obs.shape[0]
        if next(self.network.parameters()).device != obs.device:
            self.network.to(obs.device)
        if not self._kernel_domain(obs):
            # geometries the reference's nn.Sequential accepts and the kernels do not (e.g. 100x100 with a 625-input linear
            # layer): no torch evaluation in their place -- this package has no fallback path
            raise NotImplementedError(f"RandomNetworkAgent: observation {tuple(obs.shape)} is outside the HIP kernels' domain "
                                      "(H, W multiples of 8, fp32 weights, at most 65535 instances)")
        w1, w2, wl = (self.network[i].weight.detach().contiguous() for i in (0, 3, 7))
        action, output, self._ws = hip.agent_rn_forward(obs, w1, w2, wl, self.toggle_rate,
                                                        getattr(self, "_ws", None), want_output)
        action = action.reshape(instances, 1, self.action_width, self.action_height)
        return (action, output) if want_output else action


class DemoAgent(RandomAgent):
    """evaluation/submission.py:14-41."""


class SubmissionAgent(DemoAgent):
    """evaluation/submission.py:43-50."""
