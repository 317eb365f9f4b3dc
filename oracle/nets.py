"""oracle/nets.py -- CPU fp32 checker for the RND2D / AE2D forward passes.  TEST INFRASTRUCTURE.

Functional (torch.nn.functional, CPU, fp32) restatement of the reward networks of the
reference (riveSunder/carle, carle/mcl.py), evaluated in eval mode (dropout = identity):

  rnd_predictor  mcl.py:262-276   conv(1->4,3x3,p1) relu pool2 pool2 conv(4->1,3x3,p1) relu pool2
                                  flatten linear((H/8)(W/8)->16) tanh
  rnd_random     mcl.py:285-296   conv(1->2) relu pool2 pool2 conv(2->1) relu pool2 flatten linear tanh
  rnd_bonus      mcl.py:343-352   mean_k |(target - prediction)^2|            -> (N,)
  ae_predictor   mcl.py:452-467   conv(1->4) relu pool2 conv(4->2) relu pool2
                                  convT(2->1,k4,s2,p1) relu convT(1->1,k4,s2,p1) sigmoid
  ae_bonus       mcl.py:504-512   mean_{h,w} |(obs - prediction)^2|           -> (N,)

  agent_output   agents.py:78-87   RandomNetworkAgent.network: conv(1->4) relu pool2 conv(4->1) relu pool2
                                  flatten linear(dense -> AW*AH) sigmoid, no biases
  agent_action   agents.py:91-102  1.0 * (output <= toggle_rate)            -> (N, AW*AH)

Parameters are passed as a dict name -> array using the reference's state_dict names
("predictor.0.weight", "random_network.8.bias", ...), so the golden state_dicts feed it directly.

Parity status: PINNED against tests/golden/nets_64.npz, nets_256.npz and agent_net.npz (tests/test_oracle_nets.py),
tolerance 1e-6 absolute (same ATen kernels, same machine -> typically bit-identical).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def rnd_predictor(obs, sd, prefix="predictor."):
    p = lambda k: _t(sd[prefix + k]).float()
    x = F.conv2d(_t(obs), p("0.weight"), p("0.bias"), padding=1)
    x = F.max_pool2d(F.max_pool2d(F.relu(x), 2), 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, p("5.weight"), p("5.bias"), padding=1)), 2)
    return torch.tanh(F.linear(x.flatten(1), p("11.weight"), p("11.bias")))


def rnd_random(obs, sd, prefix="random_network."):
    p = lambda k: _t(sd[prefix + k]).float()
    x = F.conv2d(_t(obs), p("0.weight"), p("0.bias"), padding=1)
    x = F.max_pool2d(F.max_pool2d(F.relu(x), 2), 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, p("4.weight"), p("4.bias"), padding=1)), 2)
    return torch.tanh(F.linear(x.flatten(1), p("8.weight"), p("8.bias")))


def rnd_bonus(obs, sd):
    t, q = rnd_random(obs, sd), rnd_predictor(obs, sd)
    return torch.mean(torch.abs((t - q) ** 2), dim=[1])


def ae_predictor(obs, sd, prefix="predictor."):
    p = lambda k: _t(sd[prefix + k]).float()
    x = F.max_pool2d(F.relu(F.conv2d(_t(obs), p("0.weight"), p("0.bias"), padding=1)), 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, p("4.weight"), p("4.bias"), padding=1)), 2)
    x = F.relu(F.conv_transpose2d(x, p("8.weight"), p("8.bias"), stride=2, padding=1))
    return torch.sigmoid(F.conv_transpose2d(x, p("11.weight"), p("11.bias"), stride=2, padding=1))


def ae_bonus(obs, sd):
    o = _t(obs)
    return torch.mean(torch.abs((o - ae_predictor(o, sd)) ** 2), dim=[1, 2, 3])


def agent_output(obs, sd, prefix="network."):
    p = lambda k: _t(sd[prefix + k]).float()
    x = F.max_pool2d(F.relu(F.conv2d(_t(obs), p("0.weight"), None, padding=1)), 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, p("3.weight"), None, padding=1)), 2)
    return torch.sigmoid(F.linear(x.flatten(1), p("7.weight"), None))


def agent_action(obs, sd, toggle_rate=0.1):
    return 1.0 * (agent_output(obs, sd) <= toggle_rate)
