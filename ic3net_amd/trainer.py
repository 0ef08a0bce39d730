"""Trainer mirror — the rollout half of /root/reference/trainer.py (`Transition` :10-11, `get_episode`
:26-126, `run_batch` :227-242), batched: one call plays E = env.nenvs episodes in lock-step, every
tensor of a Transition carries a leading env dimension and everything stays on the GPU until the
per-episode statistics are reduced once at the end.

Differences forced by batching:
  * an env whose episode ended early (PP 'mixed': all predators on the prey) is frozen by the step kernel;
    its later transitions are masked through misc['alive_mask'] == 0 and misc['live'] == 0, and do not
    count in stat['num_steps'] — each env's episode is exactly the reference's episode.
  * stat values are sums over the E envs (the reference sums the same keys over episodes in merge_stat);
    run_batch adds E to stats['num_episodes'] per get_episode call.
  * `state` / `next_state` are only materialised in the Transition when args.store_states is set (the
    env reuses one (E,N,obs_dim) buffer; PP-hard is 1.19 GB per step at E = 8192).
"""
import gc
from collections import namedtuple
from collections.abc import Sequence
from inspect import signature

import torch
from torch import optim

from .action_utils import SampleClock, select_action, translate_action
from .action_utils import select_action as _select_action_default     # tests monkeypatch `select_action` (action tapes)
from . import bptt, ops
from .utils import merge_stat

Transition = namedtuple('Transition', ('state', 'action', 'action_out', 'value', 'episode_mask', 'episode_mini_mask',
                                       'next_state', 'reward', 'misc'))


class LazyEpisode(Sequence):
    """The list of Transitions get_episode returns (trainer.py:26-126), materialised on access: len(), indexing (also
    negative / slices), iteration, item assignment and `batch += episode` behave like the list they stand for."""

    def __init__(self, n, make):
        self._n, self._make, self._items = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if i not in self._items:
            self._items[i] = self._make(i)
        return self._items[i]

    def __setitem__(self, i, value):
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        self._items[i] = value

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __eq__(self, other):
        return list(self) == list(other)


class Trainer(object):
    def __init__(self, args, policy_net, env):
        self.args = args
        self.policy_net = policy_net
        self.env = env
        self.display = False
        self.last_step = False
        self.optimizer = optim.RMSprop(policy_net.parameters(), lr=args.lrate, alpha=0.97, eps=1e-6)   # trainer.py:21-22
        self.params = [p for p in self.policy_net.parameters()]
        self.clock = SampleClock(getattr(args, 'seed', 0), getattr(args, 'env_id_offset', 0))
        self.clock.episode = -1
        self.clock.env = getattr(env, 'env', None) if hasattr(getattr(env, 'env', None), '_h') else None
        self.stats = dict()
        # hipGraph capture of the per-step launch sequence (args.hip_graph): one graph per step index t, captured
        # during the second episode (the first runs eagerly as warm-up) and replayed afterwards.
        self._graphs = {}
        self._graph_pool = None
        self._graph_gen = None          # policy_net.cache_generation the graphs were captured against
        self._episodes_played = 0
        self._static = None
        self._fin_work = dict()         # scratch of ic3_episode_finalize
        self._records = None            # native update: one bptt.EpisodeRecord per episode of the batch being collected
        self._rec = None
        self._reset_takes_epoch = None
        # encoder(obs) as a sparse gather from env state (ic3_env_encode) instead of a dense obs_dim x H GEMM;
        # the dense observation is still assembled by env.step (API contract / store_states).
        if getattr(args, 'sparse_encoder', True) and hasattr(policy_net, 'obs_encoder') \
                and hasattr(getattr(env, 'env', None), 'encode'):
            policy_net.obs_encoder = env.env.encode
            if hasattr(policy_net, 'obs_table') and getattr(args, 'encoder_table', True):
                policy_net.obs_table = env.env.encode_table
            if getattr(args, 'sparse_encoder_grad', True) and hasattr(policy_net, 'obs_env'):
                policy_net.obs_env = env.env        # rollouts under autograd: gather forward + scatter backward

    # ------------------------------------------------------------------------------------------
    def get_episode(self, epoch):
        """trainer.py:26-126 for E envs in lock-step: begin_episode + max_steps x step_episode + end_episode."""
        self.begin_episode(epoch)
        check_every = int(getattr(self.args, 'done_check_every', 0))
        if self._episode_graph_ok(check_every):
            self._play_episode_graph()                              # the T step launches as ONE hipGraph replay
            return self.end_episode()
        for t in range(self.args.max_steps):
            self.step_episode(t)
            # (auto-reset: done marks an in-launch restart, the window always runs to max_steps)
            if check_every and not self._auto_reset() and (t + 1) % check_every == 0 and \
                    bool(self._buf['done'][:t + 1].to(torch.bool).any(0).all().item()):
                break                                               # trainer.py:107-108 (every env is done)
        return self.end_episode()

    def _episode_graph_ok(self, check_every=0):
        """args.hip_graph (True / 'episode'; 'step' keeps one graph per step index): get_episode replays the whole episode —
        its max_steps step launches — as ONE graph (round 4 measured: 1 % (PP-hard) to 3.6 % (TJ-medium) faster than eager
        launches, where one graph PER STEP is 1-4 % slower than eager: a graph launch costs more than the kernel launch it
        wraps).  The first episode runs eagerly (warm-up: caches, static buffers), the second is the capture."""
        a = self.args
        mode = getattr(a, 'hip_graph', False)
        if not mode or mode == 'step' or not self._use_graph() or self._episodes_played == 0 or check_every:
            return False
        raw = self.env.env
        return not self._overlap_obs() and getattr(raw, 'obs_timer', None) is None

    def _play_episode_graph(self):
        T = self.args.max_steps
        g = self._graphs.get('episode')
        if g is None:
            graph = torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            gc.collect()                                           # (no collection inside a capture: see step_episode)
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode="thread_local"):
                    for t in range(T):
                        self._step_body(t, observe=self._dense_obs())
            finally:
                if gc_was_on:
                    gc.enable()
            self._graph_gen = getattr(self.policy_net, 'cache_generation', 0)
            g = self._graphs['episode'] = dict(graph=graph, outputs=(self._state, self._info, self._prev_hid, list(self._step_out)),
                                               mega=bool(getattr(self, '_mega_last', False)))
        self.clock.t = T - 1
        g['graph'].replay()
        self._state, self._info, self._prev_hid, step_out = g['outputs']
        self._step_out = list(step_out)
        self._nsteps = T
        self._mega_last = g['mega']                                # (what the captured steps went through)

    def _auto_reset(self):
        """args.auto_reset: an env that finishes restarts inside the step launch and keeps producing transitions
        (the reference's `while: get_episode()` collection, trainer.py:227-242) instead of idling until the lock-step
        reset; one get_episode() call is then a WINDOW of max_steps steps over E streams of consecutive episodes."""
        return bool(getattr(self.args, 'auto_reset', False))

    def _stream_continues(self):
        """Collection mode (run_batch under args.auto_reset): every window after the batch's first one continues the E
        running streams of consecutive episodes."""
        st = getattr(self, '_stream', None)
        return bool(st and st.get('cont'))

    def begin_episode(self, epoch):
        args = self.args
        raw0 = getattr(self.env, 'env', None)
        if hasattr(raw0, 'set_auto_reset') and getattr(raw0, 'auto_max_steps', 0) != (args.max_steps if self._auto_reset() else 0):
            raw0.set_auto_reset(args.max_steps if self._auto_reset() else 0)
            self._graphs.clear()                                   # (a host kernel argument of the captured launches)
            self._episodes_played = min(self._episodes_played, 1)
        if bool(getattr(args, 'incremental_obs', False)) != bool(getattr(raw0, 'incremental_obs', False)) \
                and hasattr(raw0, 'set_incremental_obs'):
            raw0.set_incremental_obs(bool(getattr(args, 'incremental_obs', False)))   # experiment, see envs.py
        if raw0 is not None and hasattr(raw0, '_h'):
            # the reset obs launch is only skipped when the coming episode will take the one-launch path again (its
            # first step writes the rows of the reset state itself): the previous episode did, and nothing that decides
            # it has changed since
            # ... or nothing reads the rows at all (the training rollout of the native update: args.dense_obs off, the sparse
            # encoder reads the integer state)
            raw0.skip_reset_obs = bool(getattr(self, '_mega_last', False) and self._mega_expected(raw0)
                                       and (not self._dense_obs() or self._fused_obs()))
        self._mega_prev = bool(getattr(self, '_mega_last', False))     # (zero_hidden hands the launch its own buffers)
        self._mega_last = False
        cont = self._stream_continues()
        if self._reset_takes_epoch is None:                        # (inspect.signature costs ~0.1 ms: once)
            self._reset_takes_epoch = 'epoch' in signature(self.env.reset).parameters
        if cont:                                                   # collection mode, not the batch's first window: the E
            state = self._state                                    # streams run on (no reset; h, c, masks carried over)
        elif self._reset_takes_epoch:                              # trainer.py:28-32
            state = self.env.reset(epoch)
        else:
            state = self.env.reset()
        self._should_display = self.display and self.last_step      # trainer.py:33-36
        if self._should_display:
            self.env.display()
        E, dev = state.shape[0], state.device
        T, N, nh = args.max_steps, args.nagents, len(args.naction_heads)
        self._state = state
        self.clock.episode += 1
        lstm = bool(getattr(args, 'recurrent', False)) and getattr(args, 'rnn_type', '') == 'LSTM'
        if not cont:
            self._info = dict()
            self._prev_hid = None                                  # (set below, once the episode's buffers exist)
        self._nsteps = 0
        # Episode buffers [T, ...]: the env / sampling kernels write step t's outputs straight into slice t, so the
        # hot loop launches no bookkeeping kernels; masks and statistics are derived once in end_episode().
        # With hipGraph replay the buffers must keep their addresses, so they are allocated once and reused
        # (a Transition then stays valid until the next begin_episode).
        if self._static is None or not self._use_graph():
            z = lambda *shape, dt=torch.int32: torch.empty((T,) + shape, dtype=dt, device=dev)
            self._static = dict(action=z(nh, E, N), reward=z(E, N, dt=torch.float32), done=z(E), alive=z(E, N),
                                is_completed=z(E, N),
                                ones=torch.ones((E, N), dtype=torch.int32, device=dev),
                                zeros=torch.zeros((E, N), dtype=torch.int32, device=dev))
        self._buf = self._static
        if not cont and not lstm:
            # trainer.py:41; an LSTM policy replaces it by init_hidden() at t = 0 (trainer.py:50-51), so not allocated then.
            # Graph mode: step 0's captured launches READ this tensor on every replay — it lives in the static buffers and is
            # zeroed here, in front of the replay (a fresh torch.zeros per episode would leave the graph reading a freed block)
            if self._use_graph():
                ph = self._static.get('prev_hid')
                if ph is None or tuple(ph.shape) != (E, args.nagents, args.hid_size):
                    ph = self._static['prev_hid'] = torch.empty((E, args.nagents, args.hid_size), dtype=torch.float32, device=dev)
                ph.zero_()
                self._prev_hid = ph
            else:
                self._prev_hid = torch.zeros((E, args.nagents, args.hid_size), dtype=torch.float32, device=dev)
        self._step_out = [(None, None, None, None)] * T          # (state, action_out, value, next_state) per step
        self._rec = None
        if self._records is not None:                              # native update: what the backward pass needs later
            raw1 = self.env.env
            knet = self._kernel_net()                             # (a zero-padded twin records its own, wider, state)
            self._rec = bptt.EpisodeRecord(T, E * N, args.hid_size if knet is self.policy_net else knet.hid_size,
                                           raw1.dims.state_words, dev, recurrent=bool(getattr(args, 'recurrent', False)))
            with torch.set_grad_enabled(bool(getattr(args, 'rollout_grad', False))):   # (the grad mode the episode's steps run in)
                rec_gates = self._record_gates(knet, T, E * N)
            if rec_gates:
                try:
                    self._rec.gates = torch.empty((T, E * N, 4 * knet.hid_size), dtype=torch.float32, device=dev)
                    self._rec.xh = torch.empty((T, E * N, 2 * knet.hid_size), dtype=torch.float32, device=dev)
                except torch.cuda.OutOfMemoryError:                # (the budget above is an estimate: recompute instead)
                    self._rec.gates = self._rec.xh = None
        self._ones_comm = self._static['ones'] if args.comm_action_one else None
        self._zeros_comm = self._static['zeros']
        if self._use_graph() and self._graphs and getattr(self.policy_net, '_fc', None) is not None:
            # weights may have changed since the capture (optimizer.step, checkpoint.load): replays never call
            # forward(), so refresh the derived weight tensors here — in place, the graphs hold their addresses; if
            # they had to be re-allocated the captured graphs are stale and are dropped
            with torch.no_grad():
                self.policy_net._fused_cache()
            gen = getattr(self.policy_net, 'cache_generation', 0)
            if gen != self._graph_gen:
                self._graphs.clear()
                self._episodes_played = min(self._episodes_played, 1)   # next episode re-captures

    def _kernel_net(self):
        """The module the one-launch kernels and the native update run: the policy itself, or — for a hidden size they are
        not built for — its zero-padded twin (comm.CommNetMLP.kernel_module)."""
        km = getattr(self.policy_net, 'kernel_module', None)
        return km() if km is not None else self.policy_net

    def _record_gates(self, knet, T, R):
        """Native update: let every step launch of the recorded rollout store its cell's activated gates in the episode record
        and the inp rows (ic3_env_set_record_out), so the backward reads them instead of running the gate product again — where the record is
        the in-place one, the split gate product and its backward planes exist (hid 64 / 128), and the records of the batch stay
        within a third of the device's memory (args.record_gates=False: recompute)."""
        a = self.args
        if not getattr(a, 'record_gates', True) or not getattr(a, 'recurrent', False) or not self._rec_inplace():
            return False
        if bptt._is_baseline(self.policy_net):                    # IRIC: the stand-in's backward (bptt.standin_for_backward)
            knet = bptt.standin_for_backward(a, self.policy_net)
        if knet is None or not hasattr(knet, '_fused_cache'):
            return False
        with torch.no_grad():
            fc = knet._fused_cache()
        H = knet.hid_size
        if fc.get('ps_l_wp3') is None or fc.get('ps_l_wp3_bwd') is None or H not in (64, 128) \
                or not getattr(a, 'fused_input_grad', True):
            return False
        # what this episode's gate / inp records need against what the device can still give: free memory + the blocks torch's
        # allocator holds unused, half of it at most (the backward's own buffers, the graph pools and other ranks on the device
        # need room too); an allocation that fails anyway falls back to the recomputing backward (begin_episode)
        free, _total = torch.cuda.mem_get_info()
        cached = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
        return T * R * 6 * H * 4 <= (free + max(cached, 0)) // 2

    def _rec_inplace(self):
        """The recorded rollout of a native update reads / writes (h, c) in the episode record (no copies) when every
        step of the episode goes through ic3_policy_step with ONE communication pass."""
        raw = getattr(self.env, 'env', None)
        ok = getattr(self.policy_net, 'record_inplace_ok', None)       # (a baseline whose kernel stand-in runs zero-padded: no)
        return raw is not None and self._mega_expected(raw) and getattr(self.policy_net, 'comm_passes', 1) == 1 \
            and (ok is None or ok())

    def _mega_expected(self, raw):
        """Will step_episode go through ic3_policy_step?  (no autograd, default sampling, a policy that supports the
        env: the same conditions _step_body tests, evaluated before the episode starts)"""
        a = self.args
        if getattr(a, 'rollout_grad', False) or not a.recurrent or self.clock.env is not raw \
                or select_action is not _select_action_default:
            return False
        ok = getattr(self.policy_net, 'mega_supported', None)
        return bool(ok(raw)) if ok is not None else False

    def _dense_obs(self):
        """args.dense_obs=False skips the obs-assembly launch when nothing consumes the dense observation (sparse
        encoder active, no store_states, no autograd): env.step(..., obs=NULL) in the C ABI.  Default: assemble it."""
        a = self.args
        if getattr(a, 'dense_obs', True) or getattr(a, 'store_states', False) or getattr(a, 'rollout_grad', False):
            return True
        return getattr(self.policy_net, 'obs_encoder', None) is None

    def _use_graph(self):
        a = self.args
        # (collection windows — run_batch under args.auto_reset — step eagerly on fresh buffers: their masks / Transitions are
        #  read after later windows have played, and a step-0 graph captured in a window that starts the streams must not be
        #  replayed in one that continues them)
        return bool(getattr(a, 'hip_graph', False)) and not getattr(a, 'store_states', False) \
            and not getattr(a, 'rollout_grad', False) and self._records is None and self.clock.env is not None \
            and getattr(self, '_stream', None) is None \
            and not getattr(self, '_should_display', False) \
            and getattr(self.clock.env, 'step_timer', None) is None      # event-timed launches stay eager

    def step_episode(self, t):
        """One iteration of the hot loop trainer.py:43-108 for all E envs (eager, or as a hipGraph replay)."""
        if not self._use_graph() or self._episodes_played == 0:
            self._step_body(t, observe=self._dense_obs())
            return
        raw = self.env.env
        g = self._graphs.get(t)
        if g is None:
            # capture: the launch sequence of step t (policy kernels, sampling, env step [, obs assembly]) with the
            # buffers it reads/writes.  The obs launch stays outside the graph while it is being event-timed.
            in_graph_obs = self._dense_obs() and (not self._obs_outside_graph(raw, t) or
                                                  (self._mega_now() and self._fused_obs()))
            saved = (self._state, self._info, self._prev_hid)
            graph = torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            # thread_local: calls made by other threads (e.g. an RCCL watchdog) must not invalidate the capture
            timer, raw.obs_timer = raw.obs_timer, None        # no event records inside a capture
            # No Python garbage collection INSIDE a capture: a cycle collected there may own device resources of something
            # long dead (another Trainer's graphs, tensors, events) and releasing them (hipFree, hipGraphDestroy, ...) from
            # the capturing thread invalidates the capture — "operation failed due to a previous error during capture" at an
            # arbitrary step of a long-lived process, after which the process is not recoverable (round 4: the GPU suite at
            # step 53 / 66 of one test, depending on how much the tests before it had allocated).  torch.cuda.graph itself
            # stopped collecting in its __enter__ (torch 2.10: only under torch.compiler.config.force_cudagraph_gc), so
            # collect once in front of an episode's first capture and keep the collector off while capturing.
            if not self._graphs:
                gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode="thread_local"):
                    self._step_body(t, observe=in_graph_obs)
            finally:
                raw.obs_timer = timer
                if gc_was_on:
                    gc.enable()
            self._graph_gen = getattr(self.policy_net, 'cache_generation', 0)
            g = self._graphs[t] = dict(graph=graph, obs_inside=in_graph_obs, inputs=saved,
                                       outputs=(self._state, self._info, self._prev_hid, self._step_out[t]))
            # capture does not execute: fall through to a replay so that step t actually runs
        if g['obs_inside'] and self._obs_outside_graph(raw, t) and not (self._mega_now() and self._fused_obs()):
            # timing was switched on after capture: re-capture without the obs launch
            del self._graphs[t]
            return self.step_episode(t)
        self.clock.t = t
        g['graph'].replay()
        if not g['obs_inside'] and self._dense_obs():
            if self._overlap_obs():
                self._observe_on_side_stream(raw)
            else:
                raw.observe_timed()
        self._state, self._info, self._prev_hid, self._step_out[t] = g['outputs']
        self._nsteps = t + 1

    def _step_body(self, t, observe=True):
        args = self.args
        state, info, buf = self._state, self._info, self._buf
        store = bool(getattr(args, 'store_states', False))
        self.clock.t = t
        with torch.set_grad_enabled(bool(getattr(args, 'rollout_grad', False))):
            cont = t == 0 and self._stream_continues()            # collection mode: the window continues E running streams
            if t == 0 and args.hard_attn and args.commnet and not cont:   # trainer.py:45-46 (quirk Q22)
                info['comm_action'] = self._zeros_comm
            if torch.is_grad_enabled() and getattr(self.policy_net, 'obs_env', None) is None:
                state = state.clone()          # the env reuses its obs buffer; autograd keeps the encoder input
            if args.recurrent:                                     # trainer.py:49-60
                if args.rnn_type == 'LSTM' and t == 0 and cont:
                    if self._rec is not None and self._rec_inplace():
                        # (h, c) the previous window ended with -> slot 0 of this window's record (fresh rows are zeroed
                        # inside the launch: an env whose t == 0 starts an episode)
                        self._prev_hid = self._rec.start_from(*self._prev_hid)
                elif args.rnn_type == 'LSTM' and t == 0:
                    if self._rec is not None and not torch.is_grad_enabled() and self._rec_inplace():
                        # native update on the one-launch path: the episode record itself holds the recurrent state
                        self._prev_hid = self._rec.start()
                    elif getattr(self, '_mega_prev', False) and not torch.is_grad_enabled() \
                            and hasattr(self.policy_net, 'zero_hidden'):
                        # the previous step went through the one-launch path: hand it its own buffers, zeroed
                        self._prev_hid = self.policy_net.zero_hidden(state.shape[0], state.device)
                    else:
                        self._prev_hid = self.policy_net.init_hidden(batch_size=state.shape[0])
                fuse_draw = not torch.is_grad_enabled() and self.clock.env is not None \
                    and hasattr(self.policy_net, 'sample_into') and select_action is _select_action_default
                raw = self.env.env
                if self._rec is not None:                          # (h, c) entering step t, env state, masks
                    self._rec.record(t, self.policy_net, raw, self._prev_hid, info)
                if fuse_draw and self.clock.env is raw and getattr(self.policy_net, 'mega_ok', None) is not None \
                        and self.policy_net.mega_ok(raw, [state, self._prev_hid]):
                    return self._step_body_mega(t, observe)
                if not torch.is_grad_enabled() and self.clock.env is raw and self.clock.env is not None and not store \
                        and select_action is _select_action_default and args.rnn_type != 'LSTM' \
                        and getattr(self.policy_net, 'rnn_step_ok', None) is not None \
                        and self.policy_net.rnn_step_ok(raw, [state, self._prev_hid]):
                    return self._step_body_rnn(t, observe)         # models.RNN, tanh recurrence: one launch too (round 6)
                self._refresh_skipped_reset_obs(raw, t)
                if self._auto_reset():
                    raise NotImplementedError("args.auto_reset needs the one-launch rollout step (ic3_policy_step: "
                                              "recurrent CommNet/IC3Net, hid_size 64/128/256, no autograd): per-env "
                                              "episode starts are handled inside that launch")
                if fuse_draw:
                    self.policy_net.sample_into = (self.clock.env, buf['action'][t])
                try:
                    action_out, value, prev_hid = self.policy_net([state, self._prev_hid], info)
                finally:
                    if fuse_draw:
                        self.policy_net.sample_into = None
                if (t + 1) % args.detach_gap == 0:                 # trainer.py:56-60
                    prev_hid = (prev_hid[0].detach(), prev_hid[1].detach()) if args.rnn_type == 'LSTM' \
                        else prev_hid.detach()
                self._prev_hid = prev_hid
            else:
                raw = self.env.env
                if self._rec is not None:                          # native update: env snapshot + masks of step t
                    self._rec.record(t, self.policy_net, raw, None, info)
                if not torch.is_grad_enabled() and self.clock.env is raw and self.clock.env is not None \
                        and select_action is _select_action_default and not store \
                        and getattr(self.policy_net, 'commnet_step_ok', None) is not None \
                        and self.policy_net.commnet_step_ok(raw, state):
                    return self._step_body_commnet(t, observe)
                self._refresh_skipped_reset_obs(raw, t)
                if self._auto_reset():
                    raise NotImplementedError("args.auto_reset needs the one-launch rollout step (ic3_commnet_step: hid_size "
                                              "64/128/256, no autograd): per-env episode starts are handled inside that launch")
                action_out, value = self.policy_net(state, info)
            if getattr(self.policy_net, 'sampled', False):                               # drawn by the policy launch
                self.policy_net.sampled = False
                action = buf['action'][t]
            else:
                action = select_action(args, action_out, self.clock, out=buf['action'][t])   # trainer.py:65
            action, actual = translate_action(args, self.env, action)                    # trainer.py:66
            cur_state = state.clone() if store else None
            raw = self.env.env
            raw.out = dict(reward=buf['reward'][t], done=buf['done'][t], alive=buf['alive'][t],
                           is_completed=buf['is_completed'][t])
            try:
                if observe:
                    next_state, reward, done, info = self.env.step(actual)               # trainer.py:67
                else:
                    next_state, reward, done, info = self.env.step(actual, observe=False)
            finally:
                raw.out = None
            info = dict(info)
            if args.hard_attn and args.commnet:                    # trainer.py:70-71 (gate for the NEXT step)
                info['comm_action'] = action[-1] if not args.comm_action_one else self._ones_comm
            if getattr(self, '_should_display', False):            # trainer.py:101-102
                self.env.display()
            self._step_out[t] = (cur_state, action_out, value, next_state.clone() if store else None)
            self._state = next_state
            self._info = info
            self._nsteps = t + 1

    @staticmethod
    def _step_timer(raw):
        """The list the one-launch step appends its (start, stop, t) HIP events to, or None: `raw.step_timer` on every
        `raw.step_timer_every`-th call (bench.py: stamping both events costs ~10 us per launch, so it samples)."""
        timer = getattr(raw, 'step_timer', None)
        if timer is None:
            return None
        tick = getattr(raw, '_step_timer_tick', 0)
        raw._step_timer_tick = tick + 1
        return timer if tick % max(1, int(getattr(raw, 'step_timer_every', 1))) == 0 else None

    def _step_body_mega(self, t, observe):
        """The same iteration as _step_body through CommNetMLP.step_env (ic3_policy_step): policy forward, the action
        draws of every head and env.step are ONE launch; the obs-assembly launch follows when `observe`."""
        args, buf, state, info = self.args, self._buf, self._state, self._info
        raw = self.env.env
        store = bool(getattr(args, 'store_states', False))
        cur_state = state.clone() if store else None
        timer = self._step_timer(raw)                              # bench: HIP events around the one launch
        stamped = timer is not None and getattr(raw, 'dispatch_events', False)
        if stamped:                                                # ... stamped by the dispatch itself
            from .envs import DispatchEvent
            e0, e1 = DispatchEvent(), DispatchEvent()
            raw.set_step_events(e0, e1)
        elif timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        # next_state rows are written by the same launch (args.fused_obs, default) unless they are wanted on a second
        # stream (args.overlap_obs) — ic3_policy_step falls back to a separate obs launch by itself when the obs
        # descriptors of a tile do not fit in LDS
        fused = observe and self._fused_obs()
        rec_out = None
        if self._rec is not None and self._prev_hid[0].data_ptr() == self._rec.hs[t].data_ptr():
            rec_out = self._rec.slot(t + 1)                        # the launch writes the next slot of the episode record
        out_buf = self._static_out(t, state)
        extra = {}
        if rec_out is not None and self._rec.gates is not None:
            extra['record_out'] = (self._rec.gates[t], self._rec.xh[t])   # + the cell's gates, inp rows (bptt: no gate product later)
            self._rec.gates_n += 1
        action_out, value, prev_hid = self.policy_net.step_env(
            raw, [state, self._prev_hid], info, action=buf['action'][t], reward=buf['reward'][t], done=buf['done'][t],
            alive=buf['alive'][t], is_completed=buf['is_completed'][t], obs=raw._obs if fused else None,
            hidden_out=rec_out, out=out_buf, **extra)
        self._mega_last = True
        if timer is not None:
            if not stamped:
                e1.record(torch.cuda.current_stream())
            timer.append((e0, e1, t))
        self._prev_hid = prev_hid                                  # no autograd here: detach_gap is moot
        if observe and not fused:
            if self._overlap_obs() and not torch.cuda.is_current_stream_capturing():
                self._observe_on_side_stream(raw)                  # obs(t) on a second stream, beside step t+1
            else:
                raw.observe_timed()
        next_state = self.env._flatten_obs(raw._obs) if hasattr(self.env, '_flatten_obs') else raw._obs
        if raw.dims.kind == 2:                                     # TJ:244-247
            info = {'alive_mask': buf['alive'][t], 'is_completed': buf['is_completed'][t]}
        else:
            info = {'alive_mask_device': buf['alive'][t]}
        if args.hard_attn and args.commnet:                        # trainer.py:70-71 (gate for the NEXT step)
            info['comm_action'] = buf['action'][t][-1] if not args.comm_action_one else self._ones_comm
        if getattr(self, '_should_display', False):                # trainer.py:101-102
            self.env.display()
        self._step_out[t] = (cur_state, action_out, value, next_state.clone() if store else None)
        self._state = next_state
        self._info = info
        self._nsteps = t + 1

    def _static_out(self, t, state):
        """hipGraph mode: the launch's [log-probs | value] rows of every step index live in one static buffer, so a captured
        step allocates nothing (allocator traffic inside a capture is avoidable risk — it is what moved the failing step of
        the garbage-collection problem described in step_episode from 53 to 66).  The buffer is shared by ALL episodes
        played in graph mode: action_out / value of a Transition are views of it and are overwritten by the next episode —
        graph-mode rollouts feed statistics, never an update (_use_graph() is False while records are collected, and
        compute_grad refuses a no-grad batch).  Eager mode: None (a fresh tensor per call, a Transition keeps its action_out)."""
        if self._rec is not None and not self._use_graph():
            # native update: the rows of every step of the episode in ONE buffer of its record — compute_grad's losses then read
            # them in place (ic3_loss_gradients) instead of stacking T x heads views
            rec = self._rec
            OT = sum(int(o) for o in self.args.naction_heads) + 1
            if rec.out is None:
                rec.out = torch.empty((self.args.max_steps, state.shape[0] * self.args.nagents, OT), dtype=torch.float32,
                                      device=state.device)
            rec.out_n += 1
            return rec.out[t]
        if not self._use_graph():
            return None
        args = self.args
        OT = sum(int(o) for o in args.naction_heads) + 1
        shape = (args.max_steps, state.shape[0] * args.nagents, OT)
        ob = self._static.get('out')
        if ob is None or tuple(ob.shape) != shape:
            ob = self._static['out'] = torch.empty(shape, dtype=torch.float32, device=state.device)
        return ob[t]

    def _refresh_skipped_reset_obs(self, raw, t):
        """begin_episode skipped the reset obs launch because it EXPECTED step 0 on the one-launch path (which writes the
        rows itself); step 0 turned out to take another path: assemble the observation of the reset state now, before
        anything reads it."""
        if t == 0 and getattr(raw, 'skip_reset_obs', False) and hasattr(raw, 'observe'):
            raw.observe()
            raw.skip_reset_obs = False

    def _step_body_commnet(self, t, observe):
        """The iteration of _step_body for the NON-recurrent CommNet module through CommNetMLP.step_env_commnet
        (ic3_commnet_step): sparse encoder, communication passes, heads, the draws of every head, env.step and the dense obs
        rows of the state acted on are ONE launch."""
        args, buf, state, info = self.args, self._buf, self._state, self._info
        raw = self.env.env
        timer = self._step_timer(raw)                              # bench: HIP events stamped by the dispatch of the launch
        if timer is not None:
            from .envs import DispatchEvent
            e0, e1 = DispatchEvent(), DispatchEvent()
            raw.set_step_events(e0, e1)
            timer.append((e0, e1, t))
        action_out, value = self.policy_net.step_env_commnet(
            raw, state, info, action=buf['action'][t], reward=buf['reward'][t], done=buf['done'][t], alive=buf['alive'][t],
            is_completed=buf['is_completed'][t], obs=raw._obs if observe else None, out=self._static_out(t, state))
        self._mega_last = True                                     # (the reset obs launch may be skipped: step 0 writes the rows)
        next_state = self.env._flatten_obs(raw._obs) if hasattr(self.env, '_flatten_obs') else raw._obs
        if raw.dims.kind == 2:                                     # TJ:244-247
            info = {'alive_mask': buf['alive'][t], 'is_completed': buf['is_completed'][t]}
        else:
            info = {'alive_mask_device': buf['alive'][t]}
        if args.hard_attn and args.commnet:                        # trainer.py:70-71 (gate for the NEXT step)
            info['comm_action'] = buf['action'][t][-1] if not args.comm_action_one else self._ones_comm
        if getattr(self, '_should_display', False):                # trainer.py:101-102
            self.env.display()
        self._step_out[t] = (None, action_out, value, None)
        self._state = next_state
        self._info = info
        self._nsteps = t + 1

    def _step_body_rnn(self, t, observe):
        """The iteration of _step_body for models.RNN with the tanh recurrence (models.py:68-92, rnn_type 'MLP') through
        RNN.step_env_rnn (ic3_commnet_step with h_in): sparse encoder, h_t = tanh(affine1(obs) + affine2(h_{t-1})), heads, the draws,
        env.step and the dense obs rows of the state acted on are ONE launch.  h_t goes to one of two persistent buffers
        (the one the entering state does not live in: the same sequence of addresses in every episode, so a captured step reads /
        writes the same ones on every replay)."""
        args, buf, state, info = self.args, self._buf, self._state, self._info
        raw = self.env.env
        timer = self._step_timer(raw)
        if timer is not None:
            from .envs import DispatchEvent
            e0, e1 = DispatchEvent(), DispatchEvent()
            raw.set_step_events(e0, e1)
            timer.append((e0, e1, t))
        E, N, H = state.shape[0], args.nagents, args.hid_size
        pp = self._static.get('rnn_h')
        if pp is None or tuple(pp.shape) != (2, E, N, H):
            pp = self._static['rnn_h'] = torch.empty((2, E, N, H), dtype=torch.float32, device=state.device)
        h_out = pp[1] if self._prev_hid.data_ptr() == pp[0].data_ptr() else pp[0]
        action_out, value, h_t = self.policy_net.step_env_rnn(
            raw, [state, self._prev_hid], info, h_out, action=buf['action'][t], reward=buf['reward'][t], done=buf['done'][t],
            alive=buf['alive'][t], is_completed=buf['is_completed'][t], obs=raw._obs if observe else None,
            out=self._static_out(t, state))
        self._mega_last = True                                     # (the reset obs launch may be skipped: step 0 writes the rows)
        self._prev_hid = h_t                                       # no autograd here: detach_gap is moot
        next_state = self.env._flatten_obs(raw._obs) if hasattr(self.env, '_flatten_obs') else raw._obs
        if raw.dims.kind == 2:                                     # TJ:244-247
            info = {'alive_mask': buf['alive'][t], 'is_completed': buf['is_completed'][t]}
        else:
            info = {'alive_mask_device': buf['alive'][t]}
        if getattr(self, '_should_display', False):                # trainer.py:101-102
            self.env.display()
        self._step_out[t] = (None, action_out, value, None)
        self._state = next_state
        self._info = info
        self._nsteps = t + 1

    def _rnn_expected(self, raw):
        """Will step_episode go through ic3_commnet_step with h_in?  (models.RNN, tanh recurrence)"""
        a = self.args
        if getattr(a, 'rollout_grad', False) or not a.recurrent or a.rnn_type == 'LSTM' or self.clock.env is not raw \
                or getattr(a, 'store_states', False) or select_action is not _select_action_default:
            return False
        ok = getattr(self.policy_net, 'rnn_step_supported', None)
        return bool(ok(raw)) if ok is not None else False

    def _mega_now(self):
        """this episode (or, before its first step, the previous one) runs on the one-launch path"""
        return bool(getattr(self, '_mega_last', False) or getattr(self, '_mega_prev', False))

    def _obs_outside_graph(self, raw, t):
        """The obs-assembly launch stays outside the captured step graph when it runs on the side stream
        (args.overlap_obs) or is being event-timed (HIP events recorded in a captured graph cannot be timed; timing
        only every k-th step and keeping the other launches in their graphs was measured: no difference)."""
        return self._overlap_obs() or raw.obs_timer is not None

    def _fused_obs(self):
        """ic3_policy_step writes the rows of the state it ACTS ON; a rollout that stores states needs obs(s_{t+1}) in
        the env's buffer right after the step (Transition.next_state, and `state` of the next slot), so it keeps the
        stand-alone obs launch behind the step."""
        return bool(getattr(self.args, 'fused_obs', True)) and not self._overlap_obs() \
            and not getattr(self.args, 'store_states', False)

    def _overlap_obs(self):
        """args.overlap_obs is only honoured when nothing on the rollout path reads the dense observation (the sparse
        encoder is active): otherwise the next forward's encoder GEMM would race with the side-stream obs launch."""
        return bool(getattr(self.args, 'overlap_obs', False)) and getattr(self.policy_net, 'obs_encoder', None) is not None \
            and self.args.hid_size % 4 == 0

    def _observe_on_side_stream(self, raw):
        """args.overlap_obs: the dense observation of the new state is assembled on a second stream from a snapshot
        of the integer state (ic3_env_snapshot / ic3_env_observe_at), so this HBM-write-bound launch overlaps the
        MFMA-bound policy kernels of the next step instead of delaying them.  Nothing on the rollout path reads the
        dense observation (the encoder gathers from the state); consumers join in end_episode()."""
        main = torch.cuda.current_stream()
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=main.device)
            self._snaps = [raw.snapshot(), raw.snapshot()]
            self._snap_busy = [None, None]
            self._snap_k = 0
        k = self._snap_k
        self._snap_k ^= 1
        if self._snap_busy[k] is not None:
            main.wait_event(self._snap_busy[k])          # the launch that read this snapshot two steps ago
        raw.snapshot(out=self._snaps[k])
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            raw.observe_timed(self._snaps[k])
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._snap_busy[k] = ev

    def end_episode(self):
        """Everything get_episode derives per step in the reference (trainer.py:70-105), vectorised over the
        episode buffers: live / alive / episode masks, per-agent reward and comm-action sums, step counts."""
        args = self.args
        if getattr(self, '_side', None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)    # observations assembled on the side stream
        n, buf = self._nsteps, self._buf
        E, N = buf['reward'].shape[1:]
        has_info = self.env.env.dims.kind == 2                     # TJ reports alive_mask / is_completed in info
        graph = self._use_graph()
        # static buffers are rewritten by the next episode's replays: hand out copies (action_out / value of a
        # Transition stay graph-owned and are valid until the same step of the next episode)
        reward = buf['reward'][:n].clone() if graph else buf['reward'][:n]
        action = buf['action'][:n].clone() if graph else buf['action'][:n]
        gated = bool(args.hard_attn and args.commnet)              # trainer.py:73-75
        gate = action[:, -1] if gated and not args.comm_action_one else None
        m = self._finalize(n, buf['done'][:n], reward, buf['alive'][:n] if has_info else None,
                           buf['is_completed'][:n] if has_info else None, gate, gated and bool(args.comm_action_one))
        alive_mask, episode_mini_mask, live = m['alive_mask'], m['episode_mini_mask'], m['live']
        episode_mask = m['episode_mask'].unsqueeze(2).expand(n, E, N)                          # trainer.py:92-96
        done_t = (m['episode_mask'] == 0) if self._auto_reset() else None
        step_out = self._step_out

        def transition(t):
            cur_state, action_out, value, next_state = step_out[t]
            misc = {'alive_mask': alive_mask[t], 'live': live[t]}
            if done_t is not None:
                misc['done'] = done_t[t]                           # (E,) this transition ends its env's episode
            return Transition(cur_state, action[t], action_out, value, episode_mask[t], episode_mini_mask[t], next_state,
                              reward[t], misc)
        # The n Transition tuples (~10 tensor views each) are built when they are READ: a rollout loop that only wants the
        # statistics (bench.py, soak runs, evaluation) does not pay ~1 ms of host time per episode during which the GPU idles
        episode = LazyEpisode(n, transition)
        stat = dict()
        enemy = bool(getattr(args, 'enemy_comm', False))
        rts = None
        raw = getattr(self.env, 'env', None)
        if hasattr(self.env, 'reward_terminal') and getattr(raw, 'has_terminal_reward', True):
            rt = self.env.reward_terminal()                        # trainer.py:112-121 (PP / TJ: zeros — skipped)
            episode[-1] = episode[-1]._replace(reward=episode[-1].reward + rt)
            rts = rt.double().sum(0).cpu().numpy()
        env_stat = self.env.get_stat() if hasattr(self.env, 'get_stat') else None   # trainer.py:124-125 (synchronises)
        sums = m['read_stats']()                                   # [num_steps, zero-length envs, reward[N], gate[N]]
        num_steps = float(sums[0])
        stat['num_steps'] = num_steps                              # trainer.py:109-110
        stat['steps_taken'] = num_steps
        rsum = sums[2:2 + N]
        stat['reward'] = rsum[:args.nfriendly].copy()              # trainer.py:86
        if enemy:
            stat['enemy_reward'] = rsum[args.nfriendly:].copy()    # trainer.py:87-88
        if gated:
            csum = sums[2 + N:2 + 2 * N]
            stat['comm_action'] = csum[:args.nfriendly].copy()
            if enemy:
                stat['enemy_comm'] = csum[args.nfriendly:].copy()
        if rts is not None:
            stat['reward'] = stat['reward'] + rts[:args.nfriendly]
            if enemy:
                stat['enemy_reward'] = stat['enemy_reward'] + rts[args.nfriendly:]
        if env_stat is not None:
            if '_episodes' in env_stat:
                # auto-reset: an env that restarted on the window's last slot holds a zero-length episode: not counted
                eps = float(env_stat['_episodes'])
                n_zero = float(sums[1])
                if 'add_rate' in env_stat and eps > 0:
                    env_stat['add_rate'] = env_stat['add_rate'] * (eps - n_zero) / eps
                env_stat['_episodes'] = eps - n_zero
            merge_stat(env_stat, stat)
        self._live = m['live_after']
        self._episodes_played += 1
        if self._rec is not None:
            self._rec.finish(self._prev_hid)
            self._records.append(self._rec)
            self._rec = None
        return (episode, stat)

    def _finalize(self, n, done, reward, alive, is_completed, gate, gate_ones):
        """Masks and reduced statistics of the n slots just played: one ic3_episode_finalize launch whose fp64 sums
        reach the host with the (synchronising) env statistics read that follows; `args.fused_finalize=False` keeps
        the same derivations as separate tensor ops (the formulation the launch is tested against)."""
        args = self.args
        forced_last = n == args.max_steps                          # trainer.py:90 (auto-reset: the window's last slot
        auto = self._auto_reset()                                  # cuts the running episodes)
        if getattr(args, 'fused_finalize', True):
            m = ops.episode_finalize(done, reward, alive, is_completed, gate, gate_ones, auto, forced_last,
                                     work=self._fin_work)
            host = self._fin_work.get('host')
            if host is None or host.numel() != m['stats'].numel():
                host = self._fin_work['host'] = torch.empty(m['stats'].numel(), dtype=torch.float64).pin_memory()
            host.copy_(m['stats'], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()

            def read_stats():
                ev.synchronize()
                return host.numpy().copy()
            m['read_stats'] = read_stats
            return m
        return self._finalize_torch(n, done, reward, alive, is_completed, gate, gate_ones, auto, forced_last)

    @staticmethod
    def _finalize_torch(n, done, reward, alive, is_completed, gate, gate_ones, auto, forced_last):
        E, N = reward.shape[1:]
        dev = reward.device
        done = done.to(torch.bool)                                 # (n, E) episode_over after step t
        not_done = (~done).to(torch.float32)
        live = torch.ones((n, E), dtype=torch.float32, device=dev) # live[t] = env still running when step t starts
        if n > 1 and not auto:                                     # (auto-reset: every slot is a real transition; `done`
            live[1:] = torch.cumprod(not_done[:-1], dim=0)         #  already includes the max_steps cut of each episode)
        done_t = done.clone()
        if forced_last:
            done_t[n - 1] = True
        al = alive.to(torch.float32) if alive is not None else torch.ones_like(reward)         # trainer.py:78-81
        mini = torch.ones_like(reward)
        if is_completed is not None:                               # trainer.py:98-99 (only when not done, Q26)
            mini = torch.where(done_t.unsqueeze(2), mini, 1.0 - is_completed.to(torch.float32))
        csum = torch.zeros(N, dtype=torch.float64, device=dev)
        if gate_ones or gate is not None:
            g = torch.ones_like(reward) if gate_ones else gate.to(torch.float32)
            csum = (g * live.unsqueeze(2)).double().sum((0, 1))
        stats = torch.cat([live.double().sum().reshape(1), done[n - 1].double().sum().reshape(1),
                           reward.double().sum((0, 1)), csum])
        return dict(live=live, alive_mask=al * live.unsqueeze(2), episode_mask=(~done_t).to(torch.float32),
                    episode_mini_mask=mini, live_after=live[-1] * not_done[-1], stats=stats,
                    read_stats=lambda: stats.cpu().numpy())

    def _run_batch_streams(self, epoch):
        """run_batch in COLLECTION MODE (args.auto_reset) — trainer.py:227-242 as E parallel streams: the reference plays
        whole episodes one after the other until the batch holds batch_size steps; here every env plays whole episodes one
        after the other (an env that finishes restarts inside the step launch), window after window of max_steps slots,
        until the batch holds batch_size slots.  The reference never sees part of an episode, so what is still running
        when the batch closes is DISCARDED: a slot counts (misc['live'], alive_mask, every statistic) only if its episode
        ends inside the batch.  What the update then sees is exactly a set of complete reference episodes — per env the
        consecutive ones that fit — with episode_mask cutting the return scan at every episode end
        (trainer.py:163-175), the recurrent state and its gradient cut there too (bptt.backward_episode)."""
        args = self.args
        E, T, N = self.env.nenvs, args.max_steps, args.nagents
        self.stats = dict()
        wins, slots = [], 0
        try:
            while slots < args.batch_size:
                if args.batch_size - slots <= T * E:
                    self.last_step = True
                self._stream = dict(cont=bool(wins))
                self.begin_episode(epoch)
                for t in range(T):
                    self.step_episode(t)
                wins.append(self._end_window())
                slots += T * E
        finally:
            self._stream = None
            self.last_step = False
        done_all = torch.cat([w['done'] for w in wins]).to(torch.bool)                 # (slots, E): the step ends the env's episode
        # complete[t, e] = an episode end at or after slot t: the slot's episode ends inside the batch
        complete = torch.flip(torch.cummax(torch.flip(done_all.to(torch.int32), [0]), 0).values, [0]).to(torch.float32)
        gated = bool(args.hard_attn and args.commnet)
        rsum = torch.zeros(N, dtype=torch.float64, device=done_all.device)
        csum = torch.zeros(N, dtype=torch.float64, device=done_all.device)
        batch = []
        for k, w in enumerate(wins):
            cw = complete[k * T:(k + 1) * T]
            w['live'].mul_(cw)                                     # (in place: the window's lazy Transitions read these)
            w['alive_mask'].mul_(cw.unsqueeze(2))
            rsum += (w['reward'] * cw.unsqueeze(2)).double().sum((0, 1))
            if gated:
                g = torch.ones_like(w['reward']) if args.comm_action_one else w['gate'].to(torch.float32)
                csum += (g * cw.unsqueeze(2)).double().sum((0, 1))
            batch += w['episode']
        # the cuts of the recurrence, for the backward pass: fresh[t] = the env starts an episode at slot t; keep[t] = the
        # state leaving slot t reaches slot t + 1 of the SAME episode with its gradient (no episode end, no detach point of
        # the env's own step counter: trainer.py:56-60)
        ntot = done_all.shape[0]
        fresh = torch.ones_like(done_all)
        fresh[1:] = done_all[:-1]
        idx = torch.arange(ntot, device=done_all.device).unsqueeze(1).expand(ntot, E)
        tstep = idx - torch.cummax(torch.where(fresh, idx, torch.zeros_like(idx)), 0).values
        gap = int(getattr(args, 'detach_gap', 10000))
        keep = ~(done_all | ((tstep + 1) % gap == 0))
        if self._records is not None:
            for k, rec in enumerate(self._records):
                rec.stream = dict(fresh=fresh[k * T:(k + 1) * T], keep=keep[k * T:(k + 1) * T], first=(k == 0),
                                  last=(k == len(wins) - 1))
        nsteps = float(complete.sum().item())                     # (synchronises, like the per-episode statistics read)
        st = dict(num_steps=nsteps, steps_taken=nsteps)
        enemy = bool(getattr(args, 'enemy_comm', False))
        rs = rsum.cpu().numpy()
        st['reward'] = rs[:args.nfriendly].copy()
        if enemy:
            st['enemy_reward'] = rs[args.nfriendly:].copy()
        if gated:
            cs = csum.cpu().numpy()
            st['comm_action'] = cs[:args.nfriendly].copy()
            if enemy:
                st['enemy_comm'] = cs[args.nfriendly:].copy()
        # env statistics (env_wrappers.get_stat) of the FINISHED episodes only: the device counters of the in-launch restarts
        raw = self.env.env
        ds = raw.device_stats()
        n_ep = float(ds.auto_episodes)
        if raw.dims.kind != 1 or raw.mode != 'competitive':
            st['success'] = ds.auto_success_sum
        if raw.dims.kind == 2:
            st['add_rate'] = ds.add_rate * n_ep
        merge_stat(st, self.stats)
        self.stats['num_episodes'] = n_ep
        self.stats['num_steps'] = nsteps
        return Transition(*zip(*batch)), self.stats

    def _end_window(self):
        """end_episode of a collection window: masks of the max_steps slots just played (no cut at the window's end: the
        streams run on) — statistics and the complete-episode mask are formed over the whole batch by _run_batch_streams."""
        args = self.args
        if getattr(self, '_side', None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        n, buf = self._nsteps, self._buf
        E, N = buf['reward'].shape[1:]
        has_info = self.env.env.dims.kind == 2
        reward, action, done = buf['reward'][:n], buf['action'][:n], buf['done'][:n]
        gated = bool(args.hard_attn and args.commnet)
        gate = action[:, -1] if gated and not args.comm_action_one else None
        m = ops.episode_finalize(done, reward, buf['alive'][:n] if has_info else None,
                                 buf['is_completed'][:n] if has_info else None, gate, gated and bool(args.comm_action_one),
                                 True, False, work=self._fin_work)
        alive_mask, episode_mini_mask, live = m['alive_mask'], m['episode_mini_mask'], m['live']
        episode_mask = m['episode_mask'].unsqueeze(2).expand(n, E, N)
        done_t = m['episode_mask'] == 0
        step_out = self._step_out

        def transition(t):
            cur_state, action_out, value, next_state = step_out[t]
            return Transition(cur_state, action[t], action_out, value, episode_mask[t], episode_mini_mask[t], next_state,
                              reward[t], {'alive_mask': alive_mask[t], 'live': live[t], 'done': done_t[t]})
        self._live = m['live_after']
        self._episodes_played += 1
        if self._rec is not None:
            self._rec.finish(self._prev_hid)
            self._records.append(self._rec)
            self._rec = None
        return dict(episode=LazyEpisode(n, transition), done=done, reward=reward, gate=gate, live=live, alive_mask=alive_mask)

    def run_batch(self, epoch):                                    # trainer.py:227-242
        if self._auto_reset():
            return self._run_batch_streams(epoch)
        batch = []
        self.stats = dict()
        self.stats['num_episodes'] = 0
        nsteps = 0
        E = self.env.nenvs
        while nsteps < self.args.batch_size:
            if self.args.batch_size - nsteps <= self.args.max_steps * E:
                self.last_step = True
            episode, episode_stat = self.get_episode(epoch)
            nsteps += episode_stat['num_steps']
            n_ep = episode_stat.pop('_episodes', E)               # auto-reset: E streams can hold more than E episodes
            merge_stat(episode_stat, self.stats)
            self.stats['num_episodes'] += n_ep
            batch += episode
        self.last_step = False
        self.stats['num_steps'] = nsteps
        batch = Transition(*zip(*batch))
        return batch, self.stats

    # ------------------------------------------------------------------------------------------
    def compute_grad(self, batch):
        """trainer.py:128-225 over a batch whose transitions carry a leading env dimension.  The time axis is the
        concatenation of the batched episodes (episode_mask is 0 on every env's last step, so the reversed return
        scan restarts there exactly as in the reference's episode-after-episode batch); transitions of envs that
        were already done (misc['live'] == 0) are excluded from every sum, mean and std."""
        args = self.args
        stat = dict()
        n = args.nagents
        rewards = torch.stack(batch.reward)                                   # (T, E, N)
        T, E = rewards.shape[0], rewards.shape[1]
        episode_masks = torch.stack(batch.episode_mask)
        episode_mini_masks = torch.stack(batch.episode_mini_mask)
        actions = torch.stack(batch.action).permute(0, 2, 3, 1).long()        # (T, E, N, heads)
        values = torch.stack([v.reshape(E, n) for v in batch.value])          # (T, E, N), carries the graph
        if not values.requires_grad:
            # a no-grad batch: a hipGraph / one-launch rollout (whose static output buffers are also rewritten by the next
            # episode: action_out / value of episode k do not survive episode k + 1 in graph mode) — nothing to differentiate
            raise RuntimeError("compute_grad() needs a rollout that kept the autograd graph (args.rollout_grad, as train_batch's "
                               "autograd path sets it); a hipGraph / no-grad batch cannot be differentiated — use train_batch()")
        nheads = len(batch.action_out[0])
        log_p_a = [torch.stack([ao[k] for ao in batch.action_out]) for k in range(nheads)]    # (T, E, N, A_k)
        alive_masks = torch.stack([m['alive_mask'] for m in batch.misc])      # (T, E, N), already x live
        live = torch.stack([m['live'] for m in batch.misc]).unsqueeze(2).expand(T, E, n)      # (T, E, N)

        returns = bptt._returns(args, rewards, episode_masks, episode_mini_masks)   # trainer.py:162-171
        advantages = returns - values.detach()                                # trainer.py:173-174
        if args.normalize_rewards:                                            # trainer.py:176-177 (live entries only)
            cnt = live.sum()
            mean = (advantages * live).sum() / cnt
            var = (((advantages - mean) ** 2) * live).sum() / (cnt - 1)       # torch.std() is unbiased
            advantages = (advantages - mean) / var.sqrt()
        per_head = [lp.gather(3, actions[..., k:k + 1]).squeeze(3) for k, lp in enumerate(log_p_a)]   # utils.py:42-53
        if args.advantages_per_action:                                        # trainer.py:192-194
            action_loss = sum((-advantages * lp * alive_masks).sum() for lp in per_head)
        else:
            action_loss = (-advantages * sum(per_head) * alive_masks).sum()   # trainer.py:196-197
        stat['action_loss'] = action_loss.item()
        value_loss = ((values - returns).pow(2) * alive_masks).sum()          # trainer.py:203-206
        stat['value_loss'] = value_loss.item()
        loss = action_loss + args.value_coeff * value_loss
        entropy = 0                                                           # trainer.py:211-218 (no alive mask there)
        for lp in log_p_a:
            entropy = entropy - (lp * lp.exp() * live.unsqueeze(3)).sum()
        stat['entropy'] = entropy.item()
        if args.entr > 0:
            loss = loss - args.entr * entropy
        loss.backward()
        return stat

    def _native_update(self):
        """args.native_update (default): the update runs on a NO-GRAD rollout + an explicit backward through time
        (ic3net_amd.bptt) when the policy is the recurrent CommNet / IC3Net; otherwise through autograd."""
        raw = getattr(self.env, 'env', None)
        if not (bool(getattr(self.args, 'native_update', True)) and raw is not None and hasattr(raw, '_h')):
            return False
        knet = self._kernel_net()
        if not bptt.supported(self.args if knet is self.policy_net else knet.args, knet, raw):
            return False
        if self._auto_reset():
            # collection mode: episode cuts inside the windows — bptt.backward_episode handles them for every family; the
            # ROLLOUT must be a one-launch step (the launch restarts finished envs itself): ic3_policy_step for the recurrent
            # LSTM policies (CommNet / IC3Net with any number of passes, IRIC through its stand-in), ic3_commnet_step for the
            # non-recurrent ones (CommNet, IC through its stand-in) and — with h_in — the tanh-recurrence RNN.
            with torch.no_grad():                                  # (the native update's rollout runs without autograd)
                if getattr(self.args, 'recurrent', False):
                    return self._mega_expected(raw) if getattr(self.args, 'rnn_type', '') == 'LSTM' else self._rnn_expected(raw)
                return self._commnet_expected(raw)
        return True

    def _commnet_expected(self, raw):
        """Will step_episode go through ic3_commnet_step?  (the non-recurrent twin of _mega_expected)"""
        a = self.args
        if getattr(a, 'rollout_grad', False) or a.recurrent or self.clock.env is not raw or getattr(a, 'store_states', False) \
                or select_action is not _select_action_default:
            return False
        net = self.policy_net
        if getattr(net, 'commnet_step_ok', None) is None or not hasattr(raw, '_h'):
            return False
        H = getattr(self._kernel_net(), 'hid_size', a.hid_size)
        return bool(ops.commnet_step_supported(raw, H)) and getattr(net.obs_encoder, '__self__', None) is raw

    def compute_grad_native(self, batch, records):
        """compute_grad() without an autograd graph: losses and dL/d(logits, value) from the batch, then
        bptt.backward_episode over every recorded episode; fills p.grad like loss.backward() would."""
        stat, d_out = bptt.loss_gradients(self.args, batch, records)
        knet = self._kernel_net()
        kargs = self.args if knet is self.policy_net else knet.args
        acc = bptt.new_accumulators(knet)
        t0 = 0
        raw = self.env.env
        with torch.no_grad():
            if records and getattr(records[0], 'stream', None) is not None:
                # collection mode: the windows are consecutive slots of the same E streams — last window first, the gradient
                # of the recurrent state handed from a window's first slot to the previous window's last one
                offs = [0]
                for rec in records:
                    offs.append(offs[-1] + rec.n)
                assert offs[-1] == d_out.shape[0], "recorded steps do not match the batch"
                carry = None
                for k in reversed(range(len(records))):
                    carry = bptt.backward_episode(kargs, knet, raw, records[k], d_out[offs[k]:offs[k + 1]], acc, carry=carry)
                t0 = offs[-1]
            else:
                for rec in records:
                    bptt.backward_episode(kargs, knet, raw, rec, d_out[t0:t0 + rec.n], acc)
                    t0 += rec.n
            assert t0 == d_out.shape[0], "recorded steps do not match the batch"
            bptt.assign_grads(knet, acc)
            if knet is not self.policy_net:
                self.policy_net.unpad_grads()                      # the twin's gradients -> the policy's own parameters
        return stat

    def train_batch(self, epoch):
        """trainer.py:245-256 (+ multi_processing.py:74-98 when torch.distributed is initialised: gradients and
        stats are summed over ranks and divided by the global num_steps)."""
        from . import sharding
        native = self._native_update()
        prev = getattr(self.args, 'rollout_grad', False)
        self.args.rollout_grad = not native             # autograd path: the rollout keeps the graph, like the reference
        self._records = [] if native else None
        # The native update reads env snapshots, never the dense observation rows (the sparse encoder and its backward work
        # from the integer state): the training rollout does not assemble them unless someone asks for the states
        # (args.store_states) or for the rows themselves (args.train_dense_obs) — 1.19 GB of stores per PP-hard step
        prev_dense = getattr(self.args, 'dense_obs', True)
        if native and not getattr(self.args, 'train_dense_obs', False):
            self.args.dense_obs = False                 # (_dense_obs() keeps them when store_states / no sparse encoder)
        try:
            batch, stat = self.run_batch(epoch)
            records = self._records
        finally:
            self.args.rollout_grad = prev
            self.args.dense_obs = prev_dense
            self._records = None
            self._rec = None
        self.optimizer.zero_grad()
        try:
            s = self.compute_grad_native(batch, records) if native else self.compute_grad(batch)
        finally:
            for r in (records or []):
                r.release()
        del records
        merge_stat(s, stat)
        stat = sharding.allreduce_stats(stat)
        sharding.allreduce_grads(self.params, stat['num_steps'])             # grads /= num_steps (trainer.py:251-253)
        self.optimizer.step()
        return stat

    def state_dict(self):                                          # trainer.py:258-262
        return self.optimizer.state_dict()

    def load_state_dict(self, state):
        self.optimizer.load_state_dict(state)
