"""TEST INFRASTRUCTURE ONLY (oracle). Counter-based random stream shared by the oracle, the
golden-vector generator and (re-implemented independently in HIP) the product kernels.

The reference draws from numpy's global MT19937 (`predator_prey_env.py:174`,
`traffic_junction_env.py:375,383,618`).  A Mersenne twister cannot be keyed per environment on a
GPU, so the build *injects* the random source: every draw is

    x24 = Philox4x32-10(counter=(draw, t, episode, domain), key=(seed, env_gid))[0] >> 8
    u   = x24 / 2**24                       (exact in fp32 and fp64)

and the decision functions are integer: `int(u*n) == (x24*n) >> 24`, `u <= p  <=>  x24 <= floor(p*2**24)`.

Domains / draw indexing (the contract the kernels follow):
  DOMAIN_PP_RESET = 1 : t = 0, draw = 0,1,2,... sequential rejection sampling of distinct cells
  DOMAIN_TJ_ADD   = 2 : t = step index inside the episode, draw = 3*r + k for arrival point r,
                        k = 0 Bernoulli(add_rate), 1 dead-slot pick, 2 route pick
  DOMAIN_SAMPLE   = 3 : t = step index, draw = head * N + agent   (action sampling)
  DOMAIN_BENCH    = 4 : synthetic random actions for env-only micro-benchmarks
"""
import numpy as np

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF

DOMAIN_PP_RESET = 1
DOMAIN_TJ_ADD = 2
DOMAIN_SAMPLE = 3
DOMAIN_BENCH = 4


def philox4x32_10(ctr, key):
    """ctr: 4 ints, key: 2 ints -> 4 uint32 (Random123 Philox4x32-10)."""
    c0, c1, c2, c3 = [int(c) & MASK for c in ctr]
    k0, k1 = [int(k) & MASK for k in key]
    for r in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def philox_vec(c0, c1, c2, c3, k0, k1):
    """Vectorised numpy version (uint64 arithmetic); all args broadcastable arrays."""
    c0, c1, c2, c3, k0, k1 = [np.asarray(a, dtype=np.uint64) & np.uint64(MASK)
                              for a in np.broadcast_arrays(c0, c1, c2, c3, k0, k1)]
    m = np.uint64(MASK)
    s = np.uint64(32)
    for r in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> s, p0 & m
        hi1, lo1 = p1 >> s, p1 & m
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & m, lo1, (hi0 ^ c3 ^ k1) & m, lo0
        k0 = (k0 + np.uint64(W0)) & m
        k1 = (k1 + np.uint64(W1)) & m
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def x24(seed, env_gid, domain, episode, t, draw):
    """The 24-bit integer behind one uniform draw."""
    return philox4x32_10((draw, t, episode, domain), (seed, env_gid))[0] >> 8


def x24_vec(seed, env_gid, domain, episode, t, draw):
    return philox_vec(draw, t, episode, domain, seed, env_gid)[0] >> np.uint32(8)


def rate_threshold(p):
    """u <= p  <=>  x24 <= floor(p * 2**24) (p*2**24 is an exact fp64 scaling)."""
    import math
    return int(min(max(math.floor(float(p) * 16777216.0), -1), 16777215))


class Stream(object):
    """Per-environment view of the stream, positioned by the harness before each env call."""

    def __init__(self, seed, env_gid):
        self.seed, self.env_gid = seed, env_gid
        self.domain = self.episode = self.t = 0
        self.ndraws = 0

    def at(self, domain, episode, t):
        self.domain, self.episode, self.t = domain, episode, t

    def draw(self, d):
        self.ndraws += 1
        return x24(self.seed, self.env_gid, self.domain, self.episode, self.t, d)
