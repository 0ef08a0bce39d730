"""TEST INFRASTRUCTURE: numpy driver of tests/host/libic3rollout_host.so — the product's own .hip sources compiled for the
host behind the C ABI of include/ic3_rollout.h (`device = -1`; tests/host/Makefile, tests/host/ic3_host_abi.cpp).

The classes mirror the few methods of ic3net_amd.envs the parity tests use (reset / step / get_state / observe / encode
/ encode_backward / stats / tables), on numpy buffers instead of device tensors, so that the GPU parity tests' bodies
carry over unchanged in meaning.  Nothing under ic3net_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ASAN = os.environ.get("IC3_HOST_ASAN", "0") == "1"
HOST_DEVICE = -1
_lib = None


def host_lib():
    """Build (make) and load libic3rollout_host[_asan].so with the prototypes of ic3net_amd._lib.EXPORTS."""
    global _lib
    if _lib is None:
        from ic3net_amd import _lib as binding      # prototypes + struct layouts only; the GPU library is not loaded
        if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
            import pytest
            pytest.skip("no C++20 host compiler at /opt/rocm/lib/llvm/bin/clang++ (tests/host/Makefile)")
        so = "libic3rollout_host_asan.so" if ASAN else "libic3rollout_host.so"
        r = subprocess.run(["make", "-C", os.path.join(HERE, "host"), so], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        l = C.CDLL(os.path.join(HERE, "host", so))
        for name, (res, args) in binding.EXPORTS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc < 0:
        msg = host_lib().ic3_last_error().decode("utf-8", "replace")
        if rc == -38:
            raise NotImplementedError(msg)
        if rc == -22:
            raise ValueError(msg)
        raise RuntimeError("rc=%d: %s" % (rc, msg))
    return rc


def p(a):
    """Pointer of a numpy array (None -> NULL); the array must be C-contiguous in its last axis."""
    return None if a is None else C.c_void_p(a.ctypes.data)


class HostEnv(object):
    PP_FIELDS = ("loc_r", "loc_c", "reached", "over", "success", "episode", "t")
    TJ_FIELDS = ("alive", "wait", "loc_r", "loc_c", "last_act", "route_loc", "route_id", "is_completed", "cars_in_sys",
                 "has_failed", "over", "episode", "t")

    def __init__(self, handle, kind):
        from ic3net_amd import _lib as binding
        self.lib, self._h, self.kind = host_lib(), handle, kind
        d = binding.Dims()
        check(self.lib.ic3_env_dims(self._h, C.byref(d)))
        self.dims = d
        self.E, self.N, self.obs_dim = d.E, d.N, d.obs_dim

    @classmethod
    def pp(cls, N, dim, vision, mode, E, seed=0, offset=0, no_stay=False, enemy_comm=False, device=HOST_DEVICE):
        from ic3net_amd import _lib as binding
        cfg = binding.PPCfg(E, N, 1, dim, vision, binding.PP_MODES[mode], int(not no_stay), 0, int(enemy_comm), seed, offset)
        h = C.c_void_p()
        check(host_lib().ic3_pp_create(C.byref(cfg), device, C.byref(h)))
        return cls(h, 'pp')

    @classmethod
    def tj(cls, N, dim, vision, difficulty, E, seed=0, offset=0, add_rate_min=0.05, add_rate_max=0.05, curr_start=0,
           curr_end=0, vocab_type='bool', device=HOST_DEVICE):
        from ic3net_amd import _lib as binding
        cfg = binding.TJCfg(E, N, dim, vision, binding.TJ_DIFFICULTY[difficulty], int(vocab_type == 'scalar'),
                            float(add_rate_min), float(add_rate_max), float(curr_start), float(curr_end), seed, offset)
        h = C.c_void_p()
        check(host_lib().ic3_tj_create(C.byref(cfg), device, C.byref(h)))
        return cls(h, 'tj')

    def close(self):
        if self._h is not None:
            self.lib.ic3_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # NaN-filled outputs: whatever the kernels leave unwritten shows up in the comparisons
    def _obs_buf(self):
        return np.full((self.E, self.N, self.obs_dim), np.nan, np.float32)

    def reset(self, epoch=-1):
        obs = self._obs_buf()
        check(self.lib.ic3_env_reset(self._h, int(epoch), p(obs), None))
        return obs

    def observe(self, snap=None):
        obs = self._obs_buf()
        if snap is None:
            check(self.lib.ic3_env_observe(self._h, p(obs), None))
        else:
            check(self.lib.ic3_env_observe_at(self._h, p(snap), p(obs), None))
        return obs

    def step(self, action, with_obs=True):
        act = np.ascontiguousarray(np.asarray(action, np.int32).reshape(self.E, self.N))
        obs = self._obs_buf() if with_obs else None
        rew = np.full((self.E, self.N), np.nan, np.float32)
        done = np.full((self.E,), -1, np.int32)
        alive = np.full((self.E, self.N), -1, np.int32)
        comp = np.full((self.E, self.N), -1, np.int32)
        check(self.lib.ic3_env_step(self._h, p(act), p(obs), p(rew), p(done), p(alive), p(comp), None))
        return obs, rew, done, dict(alive_mask=alive, is_completed=comp)

    def check_actions(self):
        rc = self.lib.ic3_env_check(self._h, None)
        if rc < 0:
            raise AssertionError(self.lib.ic3_last_error().decode())

    def _field(self, name):
        off, cnt = C.c_int64(), C.c_int64()
        check(self.lib.ic3_env_state_field(self._h, name.encode(), C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def raw_state(self):
        buf = np.empty(self.dims.state_words, np.int32)
        check(self.lib.ic3_env_get_state(self._h, p(buf), buf.nbytes, None))
        return buf

    def get_state(self):
        buf = self.raw_state()
        out = {}
        for name in (self.PP_FIELDS if self.kind == 'pp' else self.TJ_FIELDS):
            off, cnt = self._field(name)
            a = buf[off:off + cnt].copy()
            out[name] = a.reshape(self.E, -1) if cnt != self.E else a
        return out

    def set_state(self, **fields):
        buf = self.raw_state()
        for name, val in fields.items():
            off, cnt = self._field(name)
            buf[off:off + cnt] = np.asarray(val, np.int32).reshape(-1)
        check(self.lib.ic3_env_set_state(self._h, p(buf), buf.nbytes, None))

    def reset_to(self, state, epoch=-1):
        """ic3_env_reset_to with a FULL state dump (dict as from get_state, or the raw int32 buffer)."""
        if isinstance(state, dict):
            buf = self.raw_state()
            for name, val in state.items():
                off, cnt = self._field(name)
                buf[off:off + cnt] = np.asarray(val, np.int32).reshape(-1)
            state = buf
        obs = self._obs_buf()
        check(self.lib.ic3_env_reset_to(self._h, int(epoch), p(state), state.nbytes, p(obs), None))
        return obs

    def snapshot(self):
        snap = np.empty(self.dims.state_words, np.int32)
        check(self.lib.ic3_env_snapshot(self._h, p(snap), None))
        return snap

    def stats(self):
        from ic3net_amd import _lib as binding
        s = binding.Stats()
        check(self.lib.ic3_env_stats(self._h, C.byref(s), None))
        return s

    @property
    def add_rate(self):
        a, e = C.c_double(), C.c_double()
        check(self.lib.ic3_tj_get_add_rate(self._h, C.byref(a), C.byref(e)))
        return a.value

    def tables(self):
        d = self.dims
        grid = np.empty((d.grid_h, d.grid_w), np.int32)
        off = np.empty(d.npath + 1, np.int32)
        n = check(self.lib.ic3_tj_get_tables(self._h, p(grid), p(off), None, 0))
        rc = np.empty(n, np.int32)
        check(self.lib.ic3_tj_get_tables(self._h, None, None, p(rc), n))
        return grid, off, rc.reshape(-1, 2)

    def encode_table(self, wt):
        H = wt.shape[1]
        table = np.full((self.dims.grid_h * self.dims.grid_w, H), np.nan, np.float32)
        check(self.lib.ic3_env_encode_table(self._h, p(wt), H, p(table), None))
        return table

    def encode(self, wt, bias, loc_table=None, snap=None, ldo=None):
        H = wt.shape[1]
        ldo = H if ldo is None else ldo
        out = np.full((self.E * self.N, ldo), np.nan, np.float32)
        check(self.lib.ic3_env_encode_at(self._h, p(snap), p(wt), p(bias), p(loc_table), p(out), ldo, H, None))
        return out[:, :H].reshape(self.E, self.N, H)

    def encode_backward(self, g, snap=None, want_bias=True):
        H = g.shape[-1]
        g2 = np.ascontiguousarray(g.reshape(-1, H), np.float32)
        n = self.lib.ic3_env_encode_backward_work(self._h, H)
        check(int(n))
        work = np.full((int(n),), np.nan, np.float32)
        dwt = np.full((self.obs_dim, H), np.nan, np.float32)
        db = np.full((H,), np.nan, np.float32) if want_bias else None
        check(self.lib.ic3_env_encode_backward(self._h, p(snap), p(g2), H, H, p(dwt), p(db), p(work), None))
        return dwt, db


class HostPolicy(object):
    """The derived weights ic3_policy_step streams (ic3net_amd.comm._fused_cache / ops._policy_struct on numpy buffers),
    from a state_dict-shaped dict of float64 arrays (the same dict oracle.policy_ref.forward takes)."""

    def __init__(self, env, params, H, head_sizes, mode_avg=True, comm_zero=False, gate_split=False, use_table=True,
                 pass_index=0, inner=False, passes=1):
        from ic3net_amd import _lib as binding
        lib = host_lib()
        # (IC3_HOST_FORCE_SPLIT=1: every policy of the process takes the split gate product)
        gate_split = gate_split or os.environ.get('IC3_HOST_FORCE_SPLIT') == '1'
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self.H, self.heads = H, [int(a) for a in head_sizes]
        self.OT = sum(self.heads) + 1
        self.wt = f32(params['encoder.weight'].T)
        self.enc_bias = f32(params['encoder.bias'] + params['C_modules.%d.bias' % pass_index])
        self.loc_table = env.encode_table(self.wt) if (use_table and env is not None) else None
        self.c_w, self.w_ih, self.w_hh = f32(params['C_modules.%d.weight' % pass_index]), f32(params['f_module.weight_ih']), f32(params['f_module.weight_hh'])
        self.c_wp = np.full((H * H,), np.nan, np.float32)
        self.l_wp = np.full((4 * H * 2 * H,), np.nan, np.float32)
        check(lib.ic3_policy_pack(p(self.c_w), p(self.w_ih), p(self.w_hh), p(self.c_wp), p(self.l_wp), H, None))
        self.b_cat = f32(params['f_module.bias_ih'] + params['f_module.bias_hh'])
        self.w_heads = f32(np.concatenate([params['heads.%d.weight' % k] for k in range(len(self.heads))] + [params['value_head.weight']], 0))
        self.b_heads = f32(np.concatenate([params['heads.%d.bias' % k] for k in range(len(self.heads))] + [params['value_head.bias']], 0))
        pol = binding.Policy()
        pol.H, pol.nheads = H, len(self.heads)
        for i, a in enumerate(self.heads):
            pol.head_sizes[i] = a
        pol.mode_avg, pol.comm_zero = int(mode_avg), int(comm_zero)
        pol.pass_index, pol.inner_pass = int(pass_index), int(inner)
        pol.enc_wt, pol.enc_bias = self.wt.ctypes.data, self.enc_bias.ctypes.data
        pol.loc_table = self.loc_table.ctypes.data if self.loc_table is not None else None
        pol.c_wp, pol.lstm_wp, pol.lstm_bias = self.c_wp.ctypes.data, self.l_wp.ctypes.data, self.b_cat.ctypes.data
        pol.head_w, pol.head_b = self.w_heads.ctypes.data, self.b_heads.ctypes.data
        if gate_split:                                          # EXPERIMENT
            self.l_wp3 = np.zeros((3 * 2 * H * 4 * H,), np.uint16)
            check(lib.ic3_policy_pack_split(p(self.w_ih), p(self.w_hh), p(self.l_wp3), H, None))
            pol.gate_split, pol.lstm_wp3 = 1, self.l_wp3.ctypes.data
        if passes >= 2:                                         # ic3_policy.npasses: every communication pass in ONE launch
            pol.npasses = passes
            self._pass_bufs = []
            for i in range(passes):
                eb = f32(params['encoder.bias'] + params['C_modules.%d.bias' % i])
                cw = f32(params['C_modules.%d.weight' % i])      # (kept: p() of a temporary would dangle)
                cwp = np.full((H * H,), np.nan, np.float32)
                scratch = np.full((4 * H * 2 * H,), np.nan, np.float32)
                check(lib.ic3_policy_pack(p(cw), p(self.w_ih), p(self.w_hh), p(cwp), p(scratch), H, None))
                self._pass_bufs.append((eb, cwp, cw))
                pol.enc_bias_pass[i], pol.c_wp_pass[i] = eb.ctypes.data, cwp.ctypes.data
        self.struct = pol

    def inner_pass(self, env, h, c, alive_in, comm_in):
        """A non-final communication pass (comm_passes > 1, ic3_policy.inner_pass): h, c only."""
        check(env.lib.ic3_policy_step(env._h, C.byref(self.struct), p(h), p(c), p(alive_in), p(comm_in), *([None] * 8)))

    def forward(self, enc, E, N, h, c, alive_in, comm_in):
        """ic3_policy_forward: the policy half for a caller-supplied enc (E*N, H) = encoder(x) + C.bias."""
        out = np.full((E * N, self.OT), np.nan, np.float32)
        check(host_lib().ic3_policy_forward(C.byref(self.struct), p(enc), E, N, p(h), p(c), p(alive_in), p(comm_in), p(out), None))
        return out

    def step(self, env, h, c, alive_in, comm_in, with_obs=True):
        """ic3_policy_step: h, c (E*N, H) updated in place.  Returns out (E*N, OT), action (heads, E, N), obs, reward, done,
        alive, is_completed."""
        E, N = env.E, env.N
        out = np.full((E * N, self.OT), np.nan, np.float32)
        action = np.full((len(self.heads), E, N), -1, np.int32)
        obs = env._obs_buf() if with_obs else None
        rew = np.full((E, N), np.nan, np.float32)
        done = np.full((E,), -1, np.int32)
        alive = np.full((E, N), -1, np.int32)
        comp = np.full((E, N), -1, np.int32)
        check(env.lib.ic3_policy_step(env._h, C.byref(self.struct), p(h), p(c), p(alive_in), p(comm_in), p(out), p(action),
                                      p(obs), p(rew), p(done), p(alive), p(comp), None))
        return out, action, obs, rew, done, alive, comp
