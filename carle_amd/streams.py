"""carle_amd.streams -- the reward stream: a second HIP stream for everything that consumes
observations / rewards of step t while the CA kernel of step t+1 already runs on the caller's
stream (the bandwidth-bound CA step and the compute/latency-bound reward work overlap).

Ordering uses ``carle_stream_fork`` (hipEventRecord + hipStreamWaitEvent inside the library: one
cheap ctypes call, no torch stream context switches).  Tensors touched on the reward stream are
kept referenced until the next cross-stream join (every ``join_every`` forks), so torch's caching
allocator cannot recycle them while the reward stream may still use them.
"""
import torch

from . import hip


class RewardStream:
    def __init__(self, device, join_every=8):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.raw = self.stream.cuda_stream
        self._ev_fork = torch.cuda.Event()
        self._ev_join = torch.cuda.Event()
        with torch.cuda.device(self.device):
            self._ev_fork.record()           # materialise the hipEvent_t handles
            self._ev_join.record()
        self._held = []
        self.join_every = join_every

    def fork(self, *tensors):
        """Reward stream waits for everything enqueued so far on the caller's current stream;
        ``tensors`` are the buffers the reward stream is about to touch."""
        if hip._raw_stream is not None:
            main_raw = hip._raw_stream(self.device.index)
        else:
            main_raw = torch.cuda.current_stream(self.device).cuda_stream
        hip.stream_fork(self._ev_fork.cuda_event, main_raw, self.raw)
        self._held.append(tensors)
        if len(self._held) >= self.join_every:
            hip.stream_fork(self._ev_join.cuda_event, self.raw, main_raw)   # bound the lag
            self._held = self._held[-1:]
        return self.raw

    def event_for_fork(self, main_raw, *tensors):
        """For library calls that perform the fork themselves (carle_env_step_f32_rnd): hands out
        the fork event, keeps ``tensors`` referenced and bounds the lag like ``fork``."""
        self._held.append(tensors)
        if len(self._held) >= self.join_every:
            hip.stream_fork(self._ev_join.cuda_event, self.raw, main_raw)
            self._held = self._held[-1:]
        return self._ev_fork.cuda_event

    def join(self):
        """The caller's current stream waits for all reward work enqueued so far."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self._held = []

    def context(self):
        """Makes the reward stream torch's current stream for a ``with`` block (entered every step
        by the asynchronous wrappers: a plain set_stream pair, cheaper than torch.cuda.stream())."""
        return _Current(self)


class _Current:
    __slots__ = ("rs", "prev")

    def __init__(self, rs):
        self.rs = rs
        self.prev = None

    def __enter__(self):
        self.prev = torch.cuda.current_stream(self.rs.device)
        torch.cuda.set_stream(self.rs.stream)
        return self.rs

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False
