"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: load fixtures captured from the
reference (tests/golden/make_golden.py) and rebuild dense observations from their sparse form."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PP_FIXTURES = ["pp_easy_mixed", "pp_easy_coop", "pp_easy_comp", "pp_medium_mixed", "pp_hard_mixed", "pp_edge_v2",
               "pp_nostay_v1", "pp_enemycomm_mixed", "pp_enemycomm_coop"]
TJ_FIXTURES = ["tj_easy_v0", "tj_easy_v1_full", "tj_medium_v0", "tj_medium_v1", "tj_hard_v0", "tj_hard_v1",
               "tj_hard9_v2", "tj_easy_curr", "tj_scalar_medium_v1", "tj_scalar_easy_v0", "tj_scalar_hard_v2"]
MODES = ["mixed", "cooperative", "competitive"]
DIFFS = ["easy", "medium", "hard"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


class SparseObs(object):
    """obs_coo rows (e, ep, t, agent, index, value); t = 0 is the reset observation."""

    def __init__(self, coo, N, obs_dim):
        self.N, self.obs_dim = N, obs_dim
        self.by_key = {}
        c = coo.astype(np.int64)
        for row, val in zip(c, coo[:, 5]):
            self.by_key.setdefault((row[0], row[1], row[2]), []).append((row[3], row[4], val))

    def dense(self, e, ep, t):
        out = np.zeros((self.N, self.obs_dim), np.float32)
        for a, i, v in self.by_key.get((e, ep, t), []):
            out[a, i] = v
        return out


def crc_of(*arrays):
    """CRC32 chain used by tests/golden/make_golden_sweep.py"""
    import zlib
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


SWEEP_RATES = [0.05, 0.3, 0.7, 1.0]
