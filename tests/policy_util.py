"""Shared by CPU/GPU policy tests: rebuild fixture inputs / weights (tests/golden/policy_*.npz)."""
import argparse

import numpy as np

from golden_util import load

POLICY_FIXTURES = ["policy_ic3net_small", "policy_ic3net_b3", "policy_commnet_rec", "policy_commnet_mlp2",
                   "policy_commnet_sum", "policy_commnet_maskzero", "policy_commnet_share",
                   "policy_commnet_initzeros", "policy_pphard_closed",
                   # BASELINE configs 3-5 at full size + the H = 64 branches, on real env observations / alive masks
                   # (make_golden_policy.py fullsize)
                   "policy_tjmedium_closed", "policy_tjhard_closed", "policy_ppscaled_closed",
                   "policy_h64_commnet_sum", "policy_h64_maskzero_b3", "policy_h64_ic3net_b3",
                   # comm_passes > 1 on the recurrent policy (make_golden_policy.py multipass): own C per pass / shared
                   "policy_h64_ic3net_p2", "policy_h128_commnet_p3share",
                   # the non-recurrent module at hid 64 / 128 (make_golden_policy.py nonrec): 2 passes, 3 shared passes, gated
                   "policy_h64_commnet_mlp2", "policy_h64_commnet_mlp3share", "policy_h128_ic3net_mlp1"]


def closed_form_weights(shapes, scale=0.05):
    out = {}
    for k, name in enumerate(sorted(shapes)):
        shp = shapes[name]
        n = int(np.prod(shp))
        i = np.arange(n, dtype=np.float64)
        out[name] = (scale * np.sin(0.37 * i + 1.3 * k) * np.cos(0.011 * i * (k + 1))).reshape(shp)
    return out


class PolicyCase(object):
    def __init__(self, name):
        fx = load(name)
        (self.N, self.obs_dim, self.H, self.steps, self.B, rec, self.comm_passes, avg, mz, ha, sh, self.nheads,
         closed) = [int(v) for v in fx["cfg"]]
        self.recurrent, self.mode_avg, self.mask_zero, self.hard_attn, self.share = bool(rec), bool(avg), bool(mz), \
            bool(ha), bool(sh)
        self.fx = fx
        if closed:
            shapes = {str(n): eval(str(s)) for n, s in zip(fx["param_names"], fx["param_shapes"])}
            self.params = closed_form_weights(shapes)
            if self.share:       # one C module under several state_dict names: load_state_dict copies them in order into
                last = 'C_modules.%d.' % (self.comm_passes - 1)          # the same tensor — the last name wins
                for k in list(self.params):
                    if k.startswith('C_module.') or k.startswith('C_modules.'):
                        self.params[k] = self.params[last + k.rsplit('.', 1)[1]]
            nz, val = fx["x_nz"], fx["x_val"]
            order = np.argsort(nz[0], kind='stable')
            self._nz, self._val = nz[:, order], val[order]
            self._start = np.searchsorted(self._nz[0], np.arange(self.steps + 1))
            self._x = None
        else:
            self.params = {k[2:]: fx[k] for k in fx.files if k.startswith("w:")}
            self._x = fx["x"]
        self.heads = [int(v) for v in fx["heads"]] if "heads" in fx.files else [5, 2][:self.nheads]
        self.state_f32 = self.recurrent and fx["h"].dtype == np.float32     # h / c stored as float32(reference fp64)

    def x_step(self, t):
        """dense (B, N, obs_dim) float64 input of step t (closed-form fixtures keep it sparse)"""
        if self._x is not None:
            return self._x[t]
        x = np.zeros((self.B, self.N, self.obs_dim))
        lo, hi = self._start[t], self._start[t + 1]
        x[tuple(self._nz[1:, lo:hi])] = self._val[lo:hi]
        return x

    def alive(self, t):
        a = self.fx["alive"][t]
        return None if a[0] < 0 else a

    def comm_action(self, t):
        return self.fx["comm_action"][t]

    def args(self):
        return argparse.Namespace(nagents=self.N, hid_size=self.H, comm_passes=self.comm_passes,
                                  recurrent=self.recurrent, continuous=False, naction_heads=self.heads,
                                  comm_mask_zero=self.mask_zero, share_weights=self.share, comm_init='uniform',
                                  hard_attn=self.hard_attn, comm_mode='avg' if self.mode_avg else 'sum',
                                  rnn_type='LSTM' if self.recurrent else 'MLP', init_std=0.2)
