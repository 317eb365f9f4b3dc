"""GPU tests of the reward-statistics kernels and the single-process RunningRewardStats path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 7, 256, 4096, 100003])
def test_batch_moments(n):
    from carle_amd import hip
    x = torch.randn(n, device="cuda") * 3 + 1.5
    got = hip.batch_moments_f32(x).cpu().numpy()
    xd = x.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, [n, xd.mean(), ((xd - xd.mean()) ** 2).sum()], rtol=1e-9, atol=1e-9)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    x2 = torch.randn(n + 3, device="cuda") - 4.0
    hip.batch_moments_f32(x, acc=acc)
    hip.batch_moments_f32(x2, acc=acc)
    both = np.concatenate([xd, x2.cpu().numpy().astype(np.float64)])
    np.testing.assert_allclose(acc.cpu().numpy(), [both.size, both.mean(), ((both - both.mean()) ** 2).sum()],
                               rtol=1e-9)


def test_running_stats_and_normalize():
    from carle_amd import dist as cd
    from carle_amd import hip
    stats = cd.RunningRewardStats("cuda:0", sync_every=3)
    chunks = [torch.randn(512, 1, device="cuda") * (i + 1) for i in range(7)]
    for c in chunks:
        stats.update(c)
    stats.flush()
    allx = torch.cat(chunks).cpu().numpy().astype(np.float64).reshape(-1)
    np.testing.assert_allclose([stats.count, stats.mean, stats.var],
                               [allx.size, allx.mean(), allx.var()], rtol=1e-10)
    y = stats.normalize(chunks[0], eps=1e-8).cpu().numpy().astype(np.float64)
    exp = (chunks[0].cpu().numpy().astype(np.float64) - allx.mean()) / np.sqrt(allx.var() + 1e-8)
    np.testing.assert_allclose(y, exp, rtol=1e-5, atol=1e-6)
    assert y.shape == (512, 1)
