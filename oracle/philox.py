"""oracle/philox.py -- Philox4x32-10 counter-based RNG in numpy.  TEST INFRASTRUCTURE.

Restates the published algorithm (Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3",
SC'11; Random123 philox4x32 with 10 rounds) that the device-side agents of carle_amd use
(carle_amd/csrc/agents.hip).  The reference's RandomAgent draws with torch.rand on the CPU
(carle/agents.py:38-40) and is not seeded, so parity for the agents is defined at the
distribution level; bit-level parity is against THIS restatement, pinned by the Random123
known-answer vectors (tests/test_oracle.py::test_philox_known_answers).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 counters, key: (2,) uint32 -> (..., 4) uint32 random words."""
    c = np.asarray(ctr, dtype=np.uint64) & MASK
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def bernoulli_f32(count, p, seed, offset):
    """The action generator of carle_amd: element 4q+j = (uniform_j(q) <= p) as 0.0/1.0 with
    counter (q_lo, q_hi, offset_lo, offset_hi), key (seed_lo, seed_hi), uniform = (x >> 8) * 2^-24."""
    assert count % 4 == 0
    q = np.arange(count // 4, dtype=np.uint64)
    ctr = np.stack([q & MASK, q >> np.uint64(32),
                    np.full_like(q, offset & 0xFFFFFFFF), np.full_like(q, (offset >> 32) & 0xFFFFFFFF)], axis=-1)
    x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u <= np.float32(p)).astype(np.float32).reshape(-1)
