"""Batched Predator-Prey / Traffic-Junction environments with the reference's gym-style API.

Mirrors /root/reference/ic3net-envs/ic3net_envs/predator_prey_env.py (PP) and traffic_junction_env.py
(TJ): `init_args(parser)`, `multi_agent_init(args)`, `reset([epoch])`, `step(action)`,
`reward_terminal()`, `stat`, `observation_space`, `action_space`, `seed()` — but one object holds
`args.nenvs` independent environments as struct-of-arrays in HBM, stepped by HIP kernels
(ic3net_amd/csrc).  Tensors returned are torch CUDA tensors with a leading env dimension:
obs (E, N, obs_dim) float32, reward (E, N) float32, done (E,) int32,
info['alive_mask'] / info['is_completed'] (E, N) int32.
"""
import ctypes as C

import numpy as np

from . import render as _render
import torch

from . import _lib, spaces
from ._lib import check, ptr, stream


class DispatchEvent(object):
    """A HIP event owned by the library (ic3_event_create): handed to `ic3_env_set_step_events`, it is stamped by the
    ic3_policy_step dispatch itself — no event-record packets in the stream around the launch."""

    def __init__(self):
        h = C.c_void_p()
        check(_lib.lib().ic3_event_create(C.byref(h)))
        self.handle = h

    def elapsed_time(self, end):             # same call shape as torch.cuda.Event.elapsed_time (ms)
        ms = C.c_float()
        check(_lib.lib().ic3_event_elapsed_ms(self.handle, end.handle, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            _lib.lib().ic3_event_destroy(self.handle)
        except Exception:
            pass


class _BatchedEnv(object):
    """Shared plumbing: handle lifetime, output buffers, state dump / injection."""

    def __init__(self):
        self._h = None
        self.stat = dict()
        self.episode_over = False
        self.obs_timer = None     # set to a list to collect (start, end) HIP events around every obs launch
        self.step_timer = None    # likewise around every one-launch policy+step (Trainer._step_body_mega)
        self.dispatch_events = False   # step_timer pairs stamped by the dispatch (DispatchEvent) instead of records
        self.out = None           # optional {'reward','done','alive','is_completed'} output tensors for step()
        self._last = None

    # --- handle ---------------------------------------------------------------------------------
    def _finish_init(self, handle, device):
        self._h = handle
        self.device = torch.device('cuda', device)
        d = _lib.Dims()
        check(_lib.lib().ic3_env_dims(self._h, C.byref(d)))
        self.dims = d
        self.nenvs, self.nagents_env, self.obs_dim = d.E, d.N, d.obs_dim
        E, N = d.E, d.N
        with torch.cuda.device(self.device):
            self._obs = torch.empty((E, N, d.obs_dim), dtype=torch.float32, device=self.device)
            self._reward = torch.empty((E, N), dtype=torch.float32, device=self.device)
            self._done = torch.empty((E,), dtype=torch.int32, device=self.device)
            self._alive = torch.empty((E, N), dtype=torch.int32, device=self.device)
            self._completed = torch.empty((E, N), dtype=torch.int32, device=self.device)

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().ic3_env_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _require(self):
        if self._h is None:
            raise _lib.IC3Error("multi_agent_init(args) has not been called")

    def _actions(self, action):
        """Accepts the reference's list/ndarray of N ints (E == 1) or an (E, N) tensor/array."""
        E, N = self.nenvs, self.nagents_env
        if not torch.is_tensor(action):
            action = torch.as_tensor(np.asarray(action).squeeze().astype(np.int32))
        if action.numel() != E * N:
            raise AssertionError("Action for each agent should be provided.")   # TJ:230
        return action.to(device=self.device, dtype=torch.int32).reshape(E, N).contiguous()

    # --- parity / debug --------------------------------------------------------------------------
    def get_state(self):
        """Full integer state as {field: ndarray} (synchronising)."""
        self._require()
        buf = np.empty(self.dims.state_words, np.int32)
        check(_lib.lib().ic3_env_get_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream()))
        out = {}
        for name, shape in self._fields():
            off, cnt = C.c_int64(), C.c_int64()
            check(_lib.lib().ic3_env_state_field(self._h, name.encode(), C.byref(off), C.byref(cnt)))
            out[name] = buf[off.value:off.value + cnt.value].reshape(shape).copy()
        return out

    def set_state(self, **fields):
        """Overwrite some state fields (golden initial states); others keep their current values."""
        self._require()
        buf = np.empty(self.dims.state_words, np.int32)
        check(_lib.lib().ic3_env_get_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream()))
        for name, val in fields.items():
            off, cnt = C.c_int64(), C.c_int64()
            check(_lib.lib().ic3_env_state_field(self._h, name.encode(), C.byref(off), C.byref(cnt)))
            buf[off.value:off.value + cnt.value] = np.asarray(val, np.int32).reshape(-1)
        check(_lib.lib().ic3_env_set_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream()))

    def reset_to(self, state, epoch=None):
        """reset() into a given initial state: the reset bookkeeping runs first (episode counter / Philox key, t = 0,
        over = 0, TJ curriculum, statistics), then the fields given in `state` — a dict as from get_state(), possibly
        partial — replace what reset() drew; fields not given keep their POST-reset values.  Returns the observation
        of that state.  (A full dump is what ic3_env_reset_to takes in one call.)"""
        self._require()
        lib = _lib.lib()
        buf = np.empty(self.dims.state_words, np.int32)
        with torch.cuda.device(self.device):
            check(lib.ic3_env_reset(self._h, -1 if epoch is None else int(epoch), None, stream()))
            check(lib.ic3_env_get_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream()))
            for name, val in state.items():
                off, cnt = C.c_int64(), C.c_int64()
                check(lib.ic3_env_state_field(self._h, name.encode(), C.byref(off), C.byref(cnt)))
                buf[off.value:off.value + cnt.value] = np.asarray(val, np.int32).reshape(-1)
            check(lib.ic3_env_set_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream()))
            check(lib.ic3_env_observe(self._h, ptr(self._obs), stream()))
        self.stat = dict()
        self.episode_over = False
        return self._obs

    def check_actions(self):
        """Synchronising form of the reference's action-range assert (PP:137 / TJ:228)."""
        rc = _lib.lib().ic3_env_check(self._h, stream())
        if rc < 0:
            raise AssertionError(_lib.lib().ic3_last_error().decode())

    def observe(self):
        self._require()
        check(_lib.lib().ic3_env_observe(self._h, ptr(self._obs), stream()))
        return self._obs

    def encode(self, weight_t, bias, out=None, loc_table=None):
        """encoder(obs(current state)) as a sparse gather (ic3_env_encode): weight_t = encoder.weight.t()
        contiguous (obs_dim, H), bias (H,) -> (E, N, H) float32.  Equals self.observe() @ weight_t + bias.
        loc_table = self.encode_table(weight_t) (valid while the weights are unchanged) makes it cheaper."""
        self._require()
        H = weight_t.shape[1]
        if weight_t.shape[0] != self.obs_dim or not weight_t.is_contiguous() or weight_t.dtype != torch.float32:
            raise ValueError("encode: weight_t must be a contiguous float32 (obs_dim, H) tensor")
        if loc_table is not None and (tuple(loc_table.shape) != (self.dims.grid_h * self.dims.grid_w, H)
                                      or not loc_table.is_contiguous() or loc_table.dtype != torch.float32):
            raise ValueError("encode: loc_table must be the contiguous float32 (grid_h*grid_w, H) tensor of encode_table")
        if out is None:
            out = torch.empty((self.nenvs, self.nagents_env, H), dtype=torch.float32, device=self.device)
        ldo = H if out.dim() == 3 else out.stride(0)     # (E*N, H) column slice of a wider buffer: row stride
        check(_lib.lib().ic3_env_encode(self._h, ptr(weight_t), ptr(bias), ptr(loc_table), ptr(out), ldo, H, stream()))
        return out

    def set_incremental_obs(self, on=True):
        """EXPERIMENT (ic3_env_set_incremental_obs): ic3_policy_step maintains the rows of the obs buffer it is handed
        (clear what the previous call painted, paint the new entries) instead of zero-filling them every step.  The
        caller promises not to write that buffer in between."""
        self._require()
        check(_lib.lib().ic3_env_set_incremental_obs(self._h, 1 if on else 0))
        self.incremental_obs = bool(on)

    def encode_at(self, snap, weight_t, bias, out=None, loc_table=None):
        """encode() for the state held in `snap` (a snapshot() tensor; None = the live state) — ic3_env_encode_at: the
        update half re-evaluates the encoder of a past step from its state snapshot."""
        self._require()
        H = weight_t.shape[1]
        if weight_t.shape[0] != self.obs_dim or not weight_t.is_contiguous() or weight_t.dtype != torch.float32:
            raise ValueError("encode_at: weight_t must be a contiguous float32 (obs_dim, H) tensor")
        if out is None:
            out = torch.empty((self.nenvs, self.nagents_env, H), dtype=torch.float32, device=self.device)
        ldo = H if out.dim() == 3 else out.stride(0)
        check(_lib.lib().ic3_env_encode_at(self._h, ptr(snap) if snap is not None else None, ptr(weight_t), ptr(bias),
                                           ptr(loc_table), ptr(out), ldo, H, stream()))
        return out

    def encode_table(self, weight_t):
        """Per-position sums of the location rows of weight_t (ic3_env_encode_table) for encode(loc_table=...)."""
        self._require()
        H = weight_t.shape[1]
        table = torch.empty((self.dims.grid_h * self.dims.grid_w, H), dtype=torch.float32, device=self.device)
        check(_lib.lib().ic3_env_encode_table(self._h, ptr(weight_t), H, ptr(table), stream()))
        return table

    def snapshot(self, out=None):
        """Device copy of the integer state (ic3_env_snapshot), the handle encode_backward needs later."""
        self._require()
        if out is None:
            out = torch.empty((self.dims.state_words,), dtype=torch.int32, device=self.device)
        check(_lib.lib().ic3_env_snapshot(self._h, ptr(out), stream()))
        return out

    def encode_backward(self, grad_out, snap=None, want_bias=True):
        """Gradient of encode() w.r.t. (weight_t, bias) for the state in `snap` (None = current state):
        returns (obs^T @ grad_out as (obs_dim, H), grad_out.sum over rows as (H,))  — ic3_env_encode_backward."""
        self._require()
        H = grad_out.shape[-1]
        g = grad_out.reshape(-1, H)
        if g.dtype != torch.float32 or g.stride(1) != 1 or g.shape[0] != self.nenvs * self.nagents_env:
            raise ValueError("encode_backward: grad_out must be float32 (E*N, H) with unit inner stride")
        key = ('encb', H)
        work = self._scratch.get(key) if hasattr(self, '_scratch') else None
        if work is None:
            if not hasattr(self, '_scratch'):
                self._scratch = {}
            n = _lib.lib().ic3_env_encode_backward_work(self._h, H)
            if n < 0:
                check(int(n))
            work = self._scratch[key] = torch.empty((n,), dtype=torch.float32, device=self.device)
        dwt = torch.empty((self.obs_dim, H), dtype=torch.float32, device=self.device)
        dbias = torch.empty((H,), dtype=torch.float32, device=self.device) if want_bias else None
        check(_lib.lib().ic3_env_encode_backward(self._h, ptr(snap) if snap is not None else None, ptr(g), g.stride(0), H,
                                                 ptr(dwt), ptr(dbias) if want_bias else None, ptr(work), stream()))
        return dwt, dbias

    def _encb_work(self, H):
        key = ('encb', H)
        if not hasattr(self, '_scratch'):
            self._scratch = {}
        work = self._scratch.get(key)
        if work is None:
            n = _lib.lib().ic3_env_encode_backward_work(self._h, H)
            if n < 0:
                check(int(n))
            work = self._scratch[key] = torch.empty((n,), dtype=torch.float32, device=self.device)
        return work

    def encode_backward_accumulate(self, grad_out, snap, first):
        """encode_backward over several states with ONE expansion at the end (ic3_env_encode_backward_accumulate): adds this
        state's share to the partial sums (writes them when `first`).  Returns False when the configuration has no
        partial-sums form — use encode_backward() per state then."""
        self._require()
        H = grad_out.shape[-1]
        g = grad_out.reshape(-1, H) if grad_out.is_contiguous() else grad_out
        if g.dtype != torch.float32 or g.dim() != 2 or g.stride(1) != 1 or g.shape[0] != self.nenvs * self.nagents_env:
            raise ValueError("encode_backward_accumulate: grad_out must be float32 (E*N, H) with unit inner stride")
        rc = _lib.lib().ic3_env_encode_backward_accumulate(self._h, ptr(snap) if snap is not None else None, ptr(g), g.stride(0),
                                                           H, ptr(self._encb_work(H)), int(bool(first)), stream())
        if rc == -38:
            return False
        check(rc)
        return True

    def encode_backward_finish(self, H, want_bias=True):
        """(dWt (obs_dim, H), dbias (H,)) of everything accumulated since the `first` encode_backward_accumulate call."""
        self._require()
        dwt = torch.empty((self.obs_dim, H), dtype=torch.float32, device=self.device)
        dbias = torch.empty((H,), dtype=torch.float32, device=self.device) if want_bias else None
        check(_lib.lib().ic3_env_encode_backward_finish(self._h, H, ptr(dwt), ptr(dbias) if want_bias else None,
                                                        ptr(self._encb_work(H)), stream()))
        return dwt, dbias

    def encode_window_work(self, H):
        """Scratch of the window form of the encoder backward (ic3_env_encode_backward_window), or None when this configuration
        has none."""
        self._require()
        key = ('encw', H)
        if not hasattr(self, '_scratch'):
            self._scratch = {}
        if key not in self._scratch:
            n = _lib.lib().ic3_env_encode_backward_window_work(self._h, H)
            if n < 0:
                check(int(n))
            self._scratch[key] = torch.empty((n,), dtype=torch.float32, device=self.device) if n > 0 else None
        return self._scratch[key]

    def encode_backward_window(self, grad_out, snaps, H, first=True):
        """Stage 1 of the encoder backward over a window of states in one launch: grad_out (T, E*N, >= H) float32 with unit inner
        stride (the rows' first H floats), snaps (T, state_words) int32."""
        self._require()
        T = grad_out.shape[0]
        if grad_out.dtype != torch.float32 or grad_out.dim() != 3 or grad_out.stride(2) != 1 or grad_out.shape[2] < H or \
                grad_out.shape[1] != self.nenvs * self.nagents_env or snaps.shape[0] < T or snaps.dtype != torch.int32:
            raise ValueError("encode_backward_window: grad_out float32 (T, E*N, >= H) with unit inner stride, snaps int32 (>= T, words)")
        check(_lib.lib().ic3_env_encode_backward_window(self._h, ptr(snaps), snaps.stride(0), T, ptr(grad_out), grad_out.stride(1),
                                                        grad_out.stride(0), H, ptr(self.encode_window_work(H)), int(bool(first)),
                                                        stream()))

    def encode_backward_window_finish(self, H, want_bias=True):
        """(dWt (obs_dim, H), dbias (H,)) of what the window form accumulated."""
        self._require()
        dwt = torch.empty((self.obs_dim, H), dtype=torch.float32, device=self.device)
        dbias = torch.empty((H,), dtype=torch.float32, device=self.device) if want_bias else None
        check(_lib.lib().ic3_env_encode_backward_window_finish(self._h, H, ptr(dwt), ptr(dbias) if want_bias else None,
                                                               ptr(self.encode_window_work(H)), stream()))
        return dwt, dbias

    def set_auto_reset(self, max_steps):
        """max_steps > 0: an env whose episode ends (episode_over, or max_steps steps played) starts its next episode
        inside the same step launch (ic3_env_set_auto_reset); 0: lock-step episodes (finished envs freeze)."""
        self._require()
        check(_lib.lib().ic3_env_set_auto_reset(self._h, int(max_steps)))
        self.auto_max_steps = int(max_steps)

    def set_record_out(self, gates, xh=None):
        """One-shot (ic3_env_set_record_out): the next ic3_policy_step also stores its cell's activated gates (R, 4H) there
        and, with `xh` (R, 2H), the inp half of its rows."""
        check(_lib.lib().ic3_env_set_record_out(self._h, ptr(gates) if gates is not None else None,
                                                ptr(xh) if xh is not None else None))

    def set_hidden_out(self, h_out, c_out):
        """One-shot (ic3_env_set_hidden_out): the next ic3_policy_step writes h', c' there instead of in place."""
        check(_lib.lib().ic3_env_set_hidden_out(self._h, ptr(h_out), ptr(c_out)))

    def set_step_events(self, start, stop):
        """Arm the next ic3_policy_step launch on this handle: the dispatch stamps `start` / `stop` (DispatchEvent)."""
        check(_lib.lib().ic3_env_set_step_events(self._h, start.handle, stop.handle))

    def device_stats(self):
        s = _lib.Stats()
        check(_lib.lib().ic3_env_stats(self._h, C.byref(s), stream()))
        return s

    def seed(self):
        return

    # --- common reset / step ---------------------------------------------------------------------
    def _reset(self, epoch):
        self._require()
        # skip_reset_obs (set by the Trainer when every rollout step writes the observation of the state it acts on —
        # ic3_policy_step with obs): the obs launch of reset() would only be overwritten by step 0's
        obs = None if getattr(self, 'skip_reset_obs', False) else self._obs
        with torch.cuda.device(self.device):
            check(_lib.lib().ic3_env_reset(self._h, -1 if epoch is None else int(epoch), ptr(obs), stream()))
        self.stat = dict()
        self.episode_over = False
        return self._obs

    def _step(self, action, observe=True):
        self._require()
        a = self._actions(action)
        # `self.out` lets a caller (the batched Trainer) receive reward / done / alive / is_completed directly in
        # slices of its own episode buffers instead of the env's per-step buffers (no copies on the hot path).
        o = self.out or {}
        reward, done = o.get('reward', self._reward), o.get('done', self._done)
        alive, completed = o.get('alive', self._alive), o.get('is_completed', self._completed)
        if self.obs_timer is None and observe:
            check(_lib.lib().ic3_env_step(self._h, ptr(a), ptr(self._obs), ptr(reward), ptr(done),
                                          ptr(alive), ptr(completed), stream()))
        else:
            check(_lib.lib().ic3_env_step(self._h, ptr(a), None, ptr(reward), ptr(done),
                                          ptr(alive), ptr(completed), stream()))
            if observe:
                self.observe_timed()
        self._last = (reward, done, alive, completed)
        return self._obs, reward, done

    def observe_timed(self, snap=None):
        """The obs-assembly launch (of the current state, or of a snapshot), bracketed by HIP events on the launch
        stream when `obs_timer` is a list."""
        sp = ptr(snap) if snap is not None else None
        if self.obs_timer is None:
            check(_lib.lib().ic3_env_observe_at(self._h, sp, ptr(self._obs), stream()))
            return self._obs
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        check(_lib.lib().ic3_env_observe_at(self._h, sp, ptr(self._obs), stream()))
        e1.record(torch.cuda.current_stream())
        self.obs_timer.append((e0, e1))
        return self._obs

    has_terminal_reward = False                    # reward_terminal() is identically zero: the Trainer skips the add

    def reward_terminal(self):
        return torch.zeros_like(self._reward)      # PP:292-293 / TJ:611-612: zeros


def _grid_to_text(cells):
    return "\n".join(" ".join(c.center(3) for c in row) for row in cells)


class PredatorPreyEnv(_BatchedEnv):
    """predator_prey_env.py:30-339, batched."""

    def __init__(self):
        super(PredatorPreyEnv, self).__init__()
        self.__version__ = "0.0.1"
        self.OUTSIDE_CLASS = 1
        self.PREY_CLASS = 2
        self.PREDATOR_CLASS = 3
        self.TIMESTEP_PENALTY = -0.05
        self.PREY_REWARD = 0
        self.POS_PREY_REWARD = 0.05

    def init_args(self, parser):            # PP:55-70 (same flags)
        env = parser.add_argument_group('Prey Predator task')
        env.add_argument('--nenemies', type=int, default=1, help="Total number of preys in play")
        env.add_argument('--dim', type=int, default=5, help="Dimension of box")
        env.add_argument('--vision', type=int, default=2, help="Vision of predator")
        env.add_argument('--moving_prey', action="store_true", default=False, help="Whether prey is fixed or moving")
        env.add_argument('--no_stay', action="store_true", default=False,
                         help="Whether predators have an action to stay in place")
        parser.add_argument('--mode', default='mixed', type=str, help='cooperative|competitive|mixed (default: mixed)')
        env.add_argument('--enemy_comm', action="store_true", default=False, help="Whether prey can communicate.")

    def multi_agent_init(self, args):       # PP:72-110
        for key in ('dim', 'vision', 'moving_prey', 'mode', 'enemy_comm'):
            setattr(self, key, getattr(args, key))
        self.nprey = args.nenemies
        self.npredator = args.nfriendly
        self.dims_grid = (self.dim, self.dim)
        self.stay = not args.no_stay
        if args.moving_prey:
            raise NotImplementedError
        if self.mode not in _lib.PP_MODES:
            raise RuntimeError("Incorrect mode, Available modes: [cooperative|competitive|mixed]")   # PP:269
        self.naction = 5 if self.stay else 4
        self.action_space = spaces.MultiDiscrete([self.naction])
        self.BASE = self.dim * self.dim
        self.OUTSIDE_CLASS += self.BASE
        self.PREY_CLASS += self.BASE
        self.PREDATOR_CLASS += self.BASE
        self.vocab_size = 1 + 1 + self.BASE + 1 + 1
        self.observation_space = spaces.Box(low=0, high=1, shape=(self.vocab_size, 2 * self.vision + 1,
                                                                   2 * self.vision + 1), dtype=int)
        device = int(getattr(args, 'device', 0) or 0)
        cfg = _lib.PPCfg(int(getattr(args, 'nenvs', 1)), self.npredator, self.nprey, self.dim, self.vision,
                         _lib.PP_MODES[self.mode], int(self.stay), int(bool(args.moving_prey)), int(bool(self.enemy_comm)),
                         int(getattr(args, 'seed', 0)) & 0xffffffff, int(getattr(args, 'env_id_offset', 0)))
        h = C.c_void_p()
        check(_lib.lib().ic3_pp_create(C.byref(cfg), device, C.byref(h)))
        self._finish_init(h, device)

    def _fields(self):
        E, N, T = self.nenvs, self.npredator, self.npredator + self.nprey
        return [("loc_r", (E, T)), ("loc_c", (E, T)), ("reached", (E, N)), ("over", (E,)), ("success", (E,)),
                ("episode", (E,)), ("t", (E,)), ("acc_success", (E,)), ("acc_episodes", (E,)), ("acc_steps", (E,))]

    def render(self, mode='human', close=False, env_index=0):
        """ONE env from a state readback, drawn the way the reference does with curses (PP:307-336): mode='cells' returns
        the draw calls `[row, x, text, color_pair]` (ic3net_amd/render.py, pinned to the reference's own drawing),
        'ansi' their text block, 'human' prints it."""
        st = self.get_state()
        calls = _render.pp_cells(st['loc_r'][env_index], st['loc_c'][env_index], self.npredator, self.dim)
        if mode == 'cells':
            return calls
        text = _render.cells_to_text(calls)
        if mode == 'human':
            print(text + "\n")
        return text

    def reset(self):                        # PP:146-168
        return self._reset(None)

    def step(self, action, observe=True):   # PP:112-144
        obs, reward, done = self._step(action, observe)
        debug = {'alive_mask_device': self._last[2]}   # PP has no alive_mask in info (trainer.py:78-81 uses ones)
        return obs, reward, done, debug


class TrafficJunctionEnv(_BatchedEnv):
    """traffic_junction_env.py:34-626, batched ('bool' vocab)."""

    def __init__(self):
        super(TrafficJunctionEnv, self).__init__()
        self.__version__ = "0.0.1"
        self.OUTSIDE_CLASS = 0
        self.ROAD_CLASS = 1
        self.CAR_CLASS = 2
        self.TIMESTEP_PENALTY = -0.01
        self.CRASH_PENALTY = -10

    def init_args(self, parser):            # TJ:60-77 (same flags)
        env = parser.add_argument_group('Traffic Junction task')
        env.add_argument('--dim', type=int, default=5, help="Dimension of box (i.e length of road) ")
        env.add_argument('--vision', type=int, default=1, help="Vision of car")
        env.add_argument('--add_rate_min', type=float, default=0.05, help="rate at which to add car (till curr. start)")
        env.add_argument('--add_rate_max', type=float, default=0.2, help=" max rate at which to add car")
        env.add_argument('--curr_start', type=float, default=0, help="start making harder after this many epochs [0]")
        env.add_argument('--curr_end', type=float, default=0, help="when to make the game hardest [0]")
        env.add_argument('--difficulty', type=str, default='easy', help="Difficulty level, easy|medium|hard")
        env.add_argument('--vocab_type', type=str, default='bool', help="Type of location vector to use, bool|scalar")

    def multi_agent_init(self, args):       # TJ:80-158
        for key in ('dim', 'vision', 'add_rate_min', 'add_rate_max', 'curr_start', 'curr_end', 'difficulty',
                    'vocab_type'):
            setattr(self, key, getattr(args, key))
        self.ncar = args.nagents
        if self.difficulty in ('medium', 'easy'):
            assert self.dim % 2 == 0, 'Only even dimension supported for now.'
            assert self.dim >= 4 + self.vision, 'Min dim: 4 + vision'
        if self.difficulty == 'hard':
            assert self.dim >= 9, 'Min dim: 9'
            assert self.dim % 3 == 0, 'Hard version works for multiple of 3. dim. only.'
        if self.vocab_type not in ('bool', 'scalar'):
            raise ValueError("vocab_type must be bool|scalar")
        self.naction = 2
        self.action_space = spaces.Discrete(self.naction)
        device = int(getattr(args, 'device', 0) or 0)
        cfg = _lib.TJCfg(int(getattr(args, 'nenvs', 1)), self.ncar, self.dim, self.vision,
                         _lib.TJ_DIFFICULTY[self.difficulty], int(self.vocab_type == 'scalar'), float(self.add_rate_min),
                         float(self.add_rate_max),
                         float(self.curr_start), float(self.curr_end), int(getattr(args, 'seed', 0)) & 0xffffffff,
                         int(getattr(args, 'env_id_offset', 0)))
        h = C.c_void_p()
        check(_lib.lib().ic3_tj_create(C.byref(cfg), device, C.byref(h)), exc=AssertionError)
        self._finish_init(h, device)
        d = self.dims
        self.dims_grid = (d.grid_h, d.grid_w)
        self.npath = d.npath
        self.vocab_size = d.vocab
        if self.vocab_type == 'bool':                          # TJ:129-138
            self.BASE = d.vocab - 3
            self.OUTSIDE_CLASS += self.BASE
            self.CAR_CLASS += self.BASE
            self.observation_space = spaces.Tuple((spaces.Discrete(self.naction), spaces.Discrete(self.npath),
                                                   spaces.MultiBinary((d.window, d.window, self.vocab_size))))
        else:                                                  # TJ:139-148
            self.observation_space = spaces.Tuple((spaces.Discrete(self.naction), spaces.Discrete(self.npath),
                                                   spaces.MultiDiscrete(self.dims_grid),
                                                   spaces.MultiBinary((d.window, d.window, self.vocab_size))))

    def _fields(self):
        E, N = self.nenvs, self.ncar
        per_agent = ["alive", "wait", "loc_r", "loc_c", "last_act", "route_loc", "route_id", "is_completed"]
        per_env = ["cars_in_sys", "has_failed", "over", "episode", "t", "acc_success", "acc_episodes", "acc_steps"]
        return [(n, (E, N)) for n in per_agent] + [(n, (E,)) for n in per_env]

    @property
    def add_rate(self):
        a, b = C.c_double(), C.c_double()
        check(_lib.lib().ic3_tj_get_add_rate(self._h, C.byref(a), C.byref(b)))
        return a.value

    def tables(self):
        """(grid (h,w), route_off (npath+1), route_rc (total,2)) — the init-time tables, host copies."""
        d = self.dims
        grid = np.empty((d.grid_h, d.grid_w), np.int32)
        off = np.empty(d.npath + 1, np.int32)
        need = check(_lib.lib().ic3_tj_get_tables(self._h, grid.ctypes.data_as(C.c_void_p),
                                                  off.ctypes.data_as(C.c_void_p), None, 0))
        rc = np.empty(need, np.int32)
        check(_lib.lib().ic3_tj_get_tables(self._h, None, None, rc.ctypes.data_as(C.c_void_p), need))
        return grid, off, rc.reshape(-1, 2)

    def render(self, mode='human', close=False, env_index=0):
        """ONE env from a state readback, drawn the way the reference does with curses (TJ:254-292, including its
        quirks: an easy-difficulty road whose id equals OUTSIDE_CLASS shows as blank, dead cars sit on the never-drawn
        cell (0, 0)): mode='cells' / 'ansi' / 'human' as for Predator-Prey."""
        st = self.get_state()
        grid, _, _ = self.tables()
        if self.vocab_type != 'bool':
            grid = np.where(grid == 1, 1, self.OUTSIDE_CLASS)      # the uploaded scalar-vocab grid holds road flags
        dead = st['alive'][env_index] == 0                         # TJ:566-567: a car that left is parked on (0, 0)
        calls = _render.tj_cells(grid.tolist(), self.OUTSIDE_CLASS, np.where(dead, 0, st['loc_r'][env_index]),
                                 np.where(dead, 0, st['loc_c'][env_index]), st['last_act'][env_index])
        if mode == 'cells':
            return calls
        text = _render.cells_to_text(calls)
        if mode == 'human':
            print(text + "\n")
        return text

    def reset(self, epoch=None):            # TJ:160-204
        return self._reset(epoch)

    def step(self, action, observe=True):   # TJ:206-252
        obs, reward, done = self._step(action, observe)
        debug = {'alive_mask': self._last[2], 'is_completed': self._last[3]}
        return obs, reward, done, debug


def tj_build_tables(dim, vision, difficulty):
    """Host-only: the Traffic-Junction init-time tables (traffic_helper.py:5-209, TJ:300-319) as computed by
    the library's C++ generator.  -> (dims, grid (h,w), route_off, route_rc (total,2)).  Needs no GPU."""
    d = _lib.Dims()
    need = _lib.lib().ic3_tj_build_tables(dim, vision, _lib.TJ_DIFFICULTY[difficulty], C.byref(d), None, None, None, 0)
    check(need, exc=AssertionError)
    grid = np.empty((d.grid_h, d.grid_w), np.int32)
    off = np.empty(d.npath + 1, np.int32)
    rc = np.empty(need, np.int32)
    check(_lib.lib().ic3_tj_build_tables(dim, vision, _lib.TJ_DIFFICULTY[difficulty], C.byref(d),
                                         grid.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                         rc.ctypes.data_as(C.c_void_p), need))
    return d, grid, off, rc.reshape(-1, 2)
