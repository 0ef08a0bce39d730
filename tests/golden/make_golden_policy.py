"""Policy-forward (F4) and Trainer-surface (F5) golden vectors from the reference's own CommNetMLP /
Trainer (fp64), see make_golden.py.  Fixtures hold weights (small configs) or a closed-form weight
recipe (full-size config), inputs and the reference's outputs."""
import os
import sys

import numpy as np
import torch

import ref_harness as rh
from oracle import philox

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('IC3_GOLDEN_OUT', HERE)   # where the fixtures are written (tests/test_golden_recipes_cpu.py: a tmp dir)


def closed_form_weights(shapes, scale=0.05):
    """Deterministic weights from the index alone (both sides regenerate them; no big blobs)."""
    out = {}
    for k, name in enumerate(sorted(shapes)):
        shp = shapes[name]
        n = int(np.prod(shp))
        i = np.arange(n, dtype=np.float64)
        out[name] = (scale * np.sin(0.37 * i + 1.3 * k) * np.cos(0.011 * i * (k + 1))).reshape(shp)
    return out


def policy_case(name, N, obs_dim, H, steps, seed, B=1, closed_form=False, **flags):
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args('predator_prey', nagents=N, hid_size=H, **flags)
    a.num_inputs = obs_dim
    heads = [5, 2] if a.hard_attn else [5]
    a.naction_heads = heads
    a.continuous = False
    a.num_actions = heads
    a.dim_actions = len(heads)
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent = True
        a.rnn_type = 'LSTM'
    torch.manual_seed(seed)
    net = ref['comm'].CommNetMLP(a, obs_dim)
    if closed_form:
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        w = closed_form_weights(shapes)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    rs = np.random.RandomState(seed)
    xs, alives, cas, outs_logp, outs_val, outs_h, outs_c = [], [], [], [[] for _ in heads], [], [], []
    hid = net.init_hidden(B) if a.recurrent else None
    for t in range(steps):
        # sparse one-hot-like observations (what the envs produce) with a few counts > 1
        x = np.zeros((B, N, obs_dim))
        for b in range(B):
            for n in range(N):
                idx = rs.choice(obs_dim, size=min(obs_dim, 11), replace=False)
                x[b, n, idx] = rs.choice([1.0, 1.0, 1.0, 2.0], size=len(idx))
        info = {}
        mode = t % 5
        if mode == 0:
            alive = None                                    # t=0: no alive_mask key (quirk Q21)
        else:
            k = [N, 0, 1, 2, max(N - 1, 0)][mode]
            alive = np.zeros(N)
            alive[rs.choice(N, size=k, replace=False)] = 1
            info['alive_mask'] = alive.copy()
        ca = (rs.rand(N) < 0.6).astype(np.int64) if t > 0 else np.zeros(N, dtype=np.int64)   # Q22
        if a.hard_attn:
            info['comm_action'] = ca
        xt = torch.from_numpy(x)
        with torch.no_grad():
            if a.recurrent:
                logp, val, hid = net([xt, hid], info)
            else:
                logp, val = net(xt, info)
        xs.append(x)
        alives.append(np.full(N, -1.0) if alive is None else alive)
        cas.append(ca)
        for k in range(len(heads)):
            outs_logp[k].append(logp[k].numpy())
        outs_val.append(val.numpy().reshape(B * N, 1))
        if a.recurrent:
            outs_h.append(hid[0].numpy())
            outs_c.append(hid[1].numpy())
    out = dict(cfg=np.array([N, obs_dim, H, steps, B, int(a.recurrent), a.comm_passes, int(a.comm_mode == 'avg'),
                             int(a.comm_mask_zero), int(bool(a.hard_attn)), int(a.share_weights), len(heads),
                             int(closed_form)], np.int32),
               x=np.array(xs), alive=np.array(alives), comm_action=np.array(cas), value=np.array(outs_val))
    if closed_form:
        # inputs are regenerated from the seed by the test (same RandomState sequence) — keep only sparse form
        nz = np.nonzero(out['x'])
        out['x_nz'] = np.stack(nz).astype(np.int32)
        out['x_val'] = out['x'][nz]
        del out['x']
    else:
        for k, v in sd.items():
            out['w:' + k] = v
    for k in range(len(heads)):
        out['logp%d' % k] = np.array(outs_logp[k])
    if a.recurrent:
        out['h'] = np.array(outs_h)
        out['c'] = np.array(outs_c)
    out['param_names'] = np.array(sorted(sd.keys()))
    out['param_shapes'] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'params', len(sd), 'steps', steps)


def fullsize_case(name, env_name, N, H, steps, seed, heads, tj=None, pp=None, gate='random', B=1, **flags):
    """F4 at BASELINE shapes (configs 3-5): the reference CommNetMLP (fp64) with closed-form weights, free-running over
    `steps` steps on REAL observations and alive masks — a Traffic-Junction / Predator-Prey rollout of the build's CPU
    oracle env (itself pinned to the reference env by the trajectory fixtures) under random actions, so the inputs have
    the sparsity and the alive patterns (n_alive = 0, 1, ...) the engine sees.  The policy's own outputs do not feed
    back into the env here (actions are random): what is pinned is comm.py:134-244 on realistic input sequences.
    Stored: sparse x, alive, comm_action, log-probs / value in fp64, h / c as float32 (file size)."""
    import oracle
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args(env_name, nagents=N, hid_size=H, **flags)
    if tj is not None:
        envs = [oracle.TJOracle(N, tj['dim'], tj['vision'], tj['difficulty'], add_rate_min=tj['add_rate'],
                                add_rate_max=tj['add_rate'], seed=seed, env_gid=900 + b) for b in range(B)]
    else:
        envs = [oracle.PPOracle(N, pp['dim'], pp['vision'], pp.get('mode', 'mixed'), seed=seed, env_gid=900 + b)
                for b in range(B)]
    obs_dim = envs[0].obs_dim
    a.num_inputs = obs_dim
    a.naction_heads = list(heads)
    a.continuous = False
    a.num_actions = list(heads)
    a.dim_actions = len(heads)
    a.recurrent = True
    a.rnn_type = 'LSTM'
    torch.manual_seed(seed)
    net = ref['comm'].CommNetMLP(a, obs_dim)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    w = closed_form_weights(shapes)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    rs = np.random.RandomState(seed)
    xs, alives, cas, outs_logp, outs_val, outs_h, outs_c = [], [], [], [[] for _ in heads], [], [], []
    hid = net.init_hidden(B)
    x = np.stack([e.reset(0) if tj is not None else e.reset() for e in envs]).astype(np.float64)
    alive = None
    for t in range(steps):
        info = {}
        if alive is not None:
            info['alive_mask'] = alive[0].copy()       # the reference shares ONE (N,) mask over the batch (comm.py:102-107)
        if t == 0:
            ca = np.zeros((B, N), np.int64)                                  # Q22
        elif gate == 'ones' or (gate == 'mixed' and t < steps // 2):
            ca = np.ones((B, N), np.int64)                                   # TJ --ic3net preset (Q27)
        else:
            ca = (rs.rand(B, N) < 0.6).astype(np.int64)
        ca[:] = ca[0]                                   # likewise one (N,) gate vector (comm.py:173)
        if a.hard_attn:
            info['comm_action'] = ca[0]
        with torch.no_grad():
            logp, val, hid = net([torch.from_numpy(x), hid], info)
        xs.append(x)
        alives.append(np.full((B, N), -1.0) if alive is None else alive.astype(np.float64))
        cas.append(ca)
        for k in range(len(heads)):
            outs_logp[k].append(logp[k].numpy())
        outs_val.append(val.numpy().reshape(B * N, 1))
        outs_h.append(hid[0].numpy().astype(np.float32))
        outs_c.append(hid[1].numpy().astype(np.float32))
        nxt, al = [], []
        for e in envs:                                                        # random env actions
            if tj is not None:
                o, _, _ = e.step((rs.rand(N) < 0.3).astype(np.int32))
                al.append(e.alive.copy())
            else:
                if not e.over.value:
                    o, _, _ = e.step(rs.randint(0, heads[0], size=N).astype(np.int32))
                else:
                    o = e.obs()
                al.append(np.ones(N))
            nxt.append(o)
        x = np.stack(nxt).astype(np.float64)
        alive = np.stack(al).astype(np.float64) if tj is not None else None
    X = np.array(xs)                                                          # (steps, B, N, obs)
    nz = np.nonzero(X)
    out = dict(cfg=np.array([N, obs_dim, H, steps, B, 1, int(a.comm_passes), int(a.comm_mode == 'avg'), int(a.comm_mask_zero),
                             int(bool(a.hard_attn)), int(bool(a.share_weights)), len(heads), 1], np.int32),
               heads=np.array(heads, np.int32), x_nz=np.stack(nz).astype(np.int32), x_val=X[nz].astype(np.float32),
               alive=np.array(alives)[:, 0], comm_action=np.array(cas)[:, 0], value=np.array(outs_val),
               h=np.array(outs_h), c=np.array(outs_c))
    for k in range(len(heads)):
        out['logp%d' % k] = np.array(outs_logp[k])
    out['param_names'] = np.array(sorted(shapes))
    out['param_shapes'] = np.array([str(shapes[k]) for k in sorted(shapes)])
    n_alive = [int(v.sum()) for v in np.array(alives).reshape(steps, -1, N)[:, 0] if v[0] >= 0]
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'obs_dim', obs_dim, 'steps', steps, 'n_alive seen', sorted(set(n_alive)),
          os.path.getsize(os.path.join(OUT, name + '.npz')) // 1024, 'KB')


def fullsize_main():
    # BASELINE config 3: TJ medium, CommNet recurrent (one head), N = 10, obs 533
    fullsize_case('policy_tjmedium_closed', 'traffic_junction', 10, 128, 40, 41, [2],
                  tj=dict(dim=14, vision=1, difficulty='medium', add_rate=0.15), commnet=True, recurrent=True)
    # BASELINE config 4: TJ hard, IC3Net, N = 20, obs 1325, 80 steps, real alive masks (0, 1, ... cars)
    fullsize_case('policy_tjhard_closed', 'traffic_junction', 20, 128, 80, 42, [2, 2],
                  tj=dict(dim=18, vision=1, difficulty='hard', add_rate=0.1), gate='mixed', ic3net=True, recurrent=True)
    # BASELINE config 5: PP scaled, IC3Net hid 256, N = 32, dim 40, vision 2 (obs 40 100)
    fullsize_case('policy_ppscaled_closed', 'predator_prey', 32, 256, 20, 43, [5, 2],
                  pp=dict(dim=40, vision=2), ic3net=True, recurrent=True)
    # the one-launch kernel's other instantiations / branches at H = 64
    fullsize_case('policy_h64_commnet_sum', 'traffic_junction', 5, 64, 24, 44, [2],
                  tj=dict(dim=6, vision=1, difficulty='easy', add_rate=0.3), commnet=True, recurrent=True,
                  comm_mode='sum')
    fullsize_case('policy_h64_maskzero_b3', 'predator_prey', 3, 64, 12, 45, [5, 2], B=3,
                  pp=dict(dim=5, vision=1), ic3net=True, recurrent=True, comm_mask_zero=True)
    fullsize_case('policy_h64_ic3net_b3', 'traffic_junction', 6, 64, 24, 46, [2, 2], B=3,
                  tj=dict(dim=6, vision=0, difficulty='easy', add_rate=0.4), ic3net=True, recurrent=True)


def multipass_main():
    """comm_passes > 1 on the recurrent policy (comm.py:179-218: every pass re-runs the communication block, its own
    C_modules[i] — or the shared one — and the SAME LSTMCell on the state the previous pass left)."""
    fullsize_case('policy_h64_ic3net_p2', 'traffic_junction', 6, 64, 24, 47, [2, 2], B=3,
                  tj=dict(dim=6, vision=0, difficulty='easy', add_rate=0.4), ic3net=True, recurrent=True, comm_passes=2)
    fullsize_case('policy_h128_commnet_p3share', 'predator_prey', 5, 128, 16, 48, [5], B=2,
                  pp=dict(dim=8, vision=1), commnet=True, recurrent=True, comm_passes=3, share_weights=True)


def nonrec_main():
    """The non-recurrent module (comm.py:127-129,220-224) at the hid sizes ic3_commnet_forward covers."""
    policy_case('policy_h64_commnet_mlp2', 4, 29, 64, 5, 31, B=2, commnet=True, comm_passes=2)
    policy_case('policy_h64_commnet_mlp3share', 5, 29, 64, 5, 32, B=3, commnet=True, comm_passes=3, share_weights=True)
    policy_case('policy_h128_ic3net_mlp1', 6, 29, 128, 6, 33, B=2, ic3net=True)


def baseline_case(name, kind, N, obs_dim, H, steps, seed, B=2, rnn_type='MLP'):
    """IC / IRIC baselines: the reference's models.MLP / models.RNN (models.py:8-97), free-running."""
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args('predator_prey', nagents=N, hid_size=H, recurrent=(kind == 'rnn'), rnn_type=rnn_type)
    a.naction_heads, a.continuous, a.num_actions, a.dim_actions = [5], False, [5], 1
    torch.manual_seed(seed)
    net = (ref['models'].RNN if kind == 'rnn' else ref['models'].MLP)(a, obs_dim)
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    rs = np.random.RandomState(seed)
    hid = None
    if kind == 'rnn':
        hid = net.init_hidden(B) if rnn_type == 'LSTM' else torch.zeros(B, N, H)
    xs, logps, vals, hs, cs = [], [], [], [], []
    for t in range(steps):
        x = (rs.rand(B, N, obs_dim) < 0.3).astype(np.float64)
        with torch.no_grad():
            if kind == 'rnn':
                logp, v, hid = net([torch.from_numpy(x), hid])
            else:
                logp, v = net(torch.from_numpy(x))
        xs.append(x)
        logps.append(logp[0].numpy())
        vals.append(v.numpy())
        if kind == 'rnn':
            hs.append((hid[0] if rnn_type == 'LSTM' else hid).numpy())
            if rnn_type == 'LSTM':
                cs.append(hid[1].numpy())
    out = dict(cfg=np.array([N, obs_dim, H, steps, B, int(kind == 'rnn'), int(rnn_type == 'LSTM')], np.int32),
               x=np.array(xs), logp0=np.array(logps), value=np.array(vals), h=np.array(hs), c=np.array(cs),
               param_names=np.array(sorted(sd)), param_shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd)]))
    for k, v in sd.items():
        out['w:' + k] = v
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, sorted(sd))


def main():
    baseline_case('baseline_mlp', 'mlp', 4, 29, 16, 4, 11)
    baseline_case('baseline_rnn', 'rnn', 4, 29, 16, 8, 12)
    baseline_case('baseline_rnn_lstm', 'rnn', 4, 29, 16, 8, 13, rnn_type='LSTM')
    policy_case('policy_ic3net_small', 4, 29, 16, 10, 1, ic3net=True, recurrent=True)
    policy_case('policy_ic3net_b3', 5, 29, 16, 6, 2, B=3, ic3net=True, recurrent=True)
    policy_case('policy_commnet_rec', 4, 29, 16, 10, 3, commnet=True, recurrent=True)
    policy_case('policy_commnet_mlp2', 4, 29, 16, 5, 4, commnet=True, comm_passes=2)
    policy_case('policy_commnet_sum', 4, 29, 16, 5, 5, commnet=True, recurrent=True, comm_mode='sum')
    policy_case('policy_commnet_maskzero', 4, 29, 16, 5, 6, commnet=True, recurrent=True, comm_mask_zero=True)
    policy_case('policy_commnet_share', 4, 29, 16, 5, 7, commnet=True, comm_passes=3, share_weights=True)
    policy_case('policy_commnet_initzeros', 4, 29, 16, 5, 8, commnet=True, recurrent=True, comm_init='zeros')
    policy_case('policy_pphard_closed', 10, 3636, 128, 80, 9, ic3net=True, recurrent=True, closed_form=True)


def trainer_case(name, env_name, T, nenv, nep, seed, greedy=False, closed_form=False, **flags):
    """F5: the reference Trainer.get_episode with `select_action` replaced by an action tape and the env RNG
    injected; one reference episode per (env, episode).  Records per-transition fields + the stat dict."""
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args(env_name, max_steps=T, seed=seed, **flags)
    env = rh.make_env(env_name, a)
    rh.finish_args(a, env)
    torch.manual_seed(seed)
    net = ref['comm'].CommNetMLP(a, a.num_inputs)
    if closed_form:       # BASELINE shapes: weights from the index alone (both sides regenerate them, no 4 MB blobs)
        sd = net.state_dict()
        cw = closed_form_weights({k: tuple(v.shape) for k, v in sd.items()})
        net.load_state_dict({k: torch.from_numpy(cw[k]) for k in sd})
    tr = ref['trainer'].Trainer(a, net, env)
    N, nh = a.nagents, len(a.naction_heads)
    rs = np.random.RandomState(seed)
    tape = np.zeros((nenv, nep, T, nh, N), np.int64)
    for h, A in enumerate(a.naction_heads):
        tape[:, :, :, h] = rs.randint(0, A, size=(nenv, nep, T, N))
    if env_name == 'traffic_junction':
        tape[:, :, :, 0] = (rs.rand(nenv, nep, T, N) < 0.3)
    rec = dict(action=np.zeros((nenv, nep, T, nh, N), np.int32), reward=np.zeros((nenv, nep, T, N)),
               episode_mask=np.zeros((nenv, nep, T, N)), episode_mini_mask=np.zeros((nenv, nep, T, N)),
               alive_mask=np.zeros((nenv, nep, T, N)), nsteps=np.zeros((nenv, nep), np.int32))
    stat_keys = None
    stats = {}
    cursor = {}
    trmod = ref['trainer']

    def taped_select(args, action_out):
        e, ep, t = cursor['e'], cursor['ep'], cursor['t']
        cursor['t'] += 1
        if env_name == 'traffic_junction':
            ref['rnd'].begin(cursor['st'], philox.DOMAIN_TJ_ADD, ep, t)
        act = tape[e, ep, t].copy()
        if greedy and env_name == 'predator_prey' and e == 0:
            raw = env.env
            for i in range(N):
                dr = raw.prey_loc[0][0] - raw.predator_loc[i][0]
                dc = raw.prey_loc[0][1] - raw.predator_loc[i][1]
                act[0, i] = (2 if dr > 0 else 0) if dr != 0 else ((1 if dc > 0 else 3) if dc != 0 else 4)
            tape[e, ep, t] = act
        return torch.from_numpy(act).view(nh, 1, N, 1)
    trmod.select_action = taped_select
    for e in range(nenv):
        st = philox.Stream(seed, 300 + e)
        env = rh.make_env(env_name, a)
        tr.env = env
        for ep in range(nep):
            cursor.update(e=e, ep=ep, t=0, st=st)
            ref['rnd'].begin(st, philox.DOMAIN_PP_RESET, ep, 0)
            episode, stat = tr.get_episode(ep)
            n = len(episode)
            rec['nsteps'][e, ep] = n
            for t, trn in enumerate(episode):
                rec['action'][e, ep, t] = np.array([np.asarray(x) for x in trn.action])
                rec['reward'][e, ep, t] = trn.reward
                rec['episode_mask'][e, ep, t] = trn.episode_mask
                rec['episode_mini_mask'][e, ep, t] = trn.episode_mini_mask
                rec['alive_mask'][e, ep, t] = trn.misc['alive_mask']
            for k, v in stat.items():
                stats.setdefault(k, np.zeros((nenv, nep) + np.shape(v)))[e, ep] = v
    trmod.select_action = ref['action_utils'].select_action
    out = dict(rec)
    out['tape'] = tape.astype(np.int32)
    for k, v in stats.items():
        out['stat:' + k] = v
    out['cfg'] = np.array([N, T, nenv, nep, nh, seed], np.int32)
    out['flags'] = np.array(repr(sorted(flags.items())))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'nsteps', rec['nsteps'].tolist(), 'stat keys', sorted(stats))


def grad_case(name, env_name, T, nenv, nep, seed, closed_form=False, model=None, **flags):
    """F5b: the reference run_batch + compute_grad (trainer.py:128-225) over nenv*nep taped episodes played one
    after the other in ONE batch (what a single reference process does), fp64.  Records the loss terms and every
    parameter gradient (before the /num_steps of train_batch), plus the weights and the action tape."""
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args(env_name, max_steps=T, seed=seed, **flags)
    env = rh.make_env(env_name, a)
    rh.finish_args(a, env)
    torch.manual_seed(seed)
    if model == 'mlp':        # the reference's IC / IRIC baselines (models.py:8-97; main.py:161-168 builds them like this)
        net = ref['models'].MLP(a, a.num_inputs)
    elif model == 'rnn':
        net = ref['models'].RNN(a, a.num_inputs)
    else:
        net = ref['comm'].CommNetMLP(a, a.num_inputs)
    if closed_form:       # BASELINE shapes: weights from the index alone (both sides regenerate them, no 4 MB blobs)
        sd = net.state_dict()
        cw = closed_form_weights({k: tuple(v.shape) for k, v in sd.items()})
        net.load_state_dict({k: torch.from_numpy(cw[k]) for k in sd})
    tr = ref['trainer'].Trainer(a, net, env)
    N, nh = a.nagents, len(a.naction_heads)
    rs = np.random.RandomState(seed)
    tape = np.zeros((nenv, nep, T, nh, N), np.int64)
    for h, A in enumerate(a.naction_heads):
        tape[:, :, :, h] = rs.randint(0, A, size=(nenv, nep, T, N))
    if env_name == 'traffic_junction':
        tape[:, :, :, 0] = (rs.rand(nenv, nep, T, N) < 0.3)
    cursor = {}
    trmod = ref['trainer']

    def taped_select(args, action_out):
        e, ep, t = cursor['e'], cursor['ep'], cursor['t']
        cursor['t'] += 1
        if env_name == 'traffic_junction':
            ref['rnd'].begin(cursor['st'], philox.DOMAIN_TJ_ADD, ep, t)
        act = tape[e, ep, t].copy()
        if env_name == 'predator_prey' and e == 0:       # steer env 0 onto the prey: early termination in the batch
            raw = tr.env.env
            for i in range(N):
                dr = raw.prey_loc[0][0] - raw.predator_loc[i][0]
                dc = raw.prey_loc[0][1] - raw.predator_loc[i][1]
                act[0, i] = (2 if dr > 0 else 0) if dr != 0 else ((1 if dc > 0 else 3) if dc != 0 else 4)
            tape[e, ep, t] = act
        return torch.from_numpy(act).view(nh, 1, N, 1)
    trmod.select_action = taped_select
    batch = []
    stats = dict(num_episodes=0)
    nsteps = np.zeros((nenv, nep), np.int32)
    streams = [philox.Stream(seed, 400 + e) for e in range(nenv)]
    envs = [rh.make_env(env_name, a) for e in range(nenv)]
    for ep in range(nep):                                # episode-major order == the batched engine's order
        for e in range(nenv):
            tr.env = envs[e]
            cursor.update(e=e, ep=ep, t=0, st=streams[e])
            ref['rnd'].begin(streams[e], philox.DOMAIN_PP_RESET, ep, 0)
            episode, stat = tr.get_episode(ep)
            nsteps[e, ep] = len(episode)
            ref['utils'].merge_stat(stat, stats)
            stats['num_episodes'] += 1
            batch += episode
    trmod.select_action = ref['action_utils'].select_action
    stats['num_steps'] = len(batch)
    batch = trmod.Transition(*zip(*batch))
    tr.optimizer.zero_grad()
    s = tr.compute_grad(batch)
    out = dict(tape=tape.astype(np.int32), nsteps=nsteps, cfg=np.array([N, T, nenv, nep, nh, seed], np.int32),
               flags=np.array(repr(sorted(flags.items()))), action_loss=s['action_loss'], value_loss=s['value_loss'],
               entropy=s.get('entropy', 0.0), num_steps=stats['num_steps'])
    if model:
        out['model'] = np.array(model)
    if closed_form:
        out['param_names'] = np.array(list(net.state_dict().keys()))
        out['param_shapes'] = np.array([repr(tuple(v.shape)) for v in net.state_dict().values()])
    else:
        for k, v in net.state_dict().items():
            out['w:' + k] = v.detach().numpy().copy()
    for k, p in net.named_parameters():
        g = np.zeros(0) if p.grad is None else p.grad.detach().numpy().copy()
        # (full-size fixtures keep the gradients as float32: the test's bar is 3e-4 of the largest entry)
        out['g:' + k] = g.astype(np.float32) if closed_form else g
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'steps', stats['num_steps'], 'losses', s)


def grad_stream_case(name, env_name, T, nenv, nwin, seed, closed_form=False, model=None, **flags):
    """F5c — collection mode: the reference's run_batch + compute_grad (trainer.py:227-242,128-225) over, per env, the
    consecutive WHOLE episodes that fit in nwin * T slots (`while: get_episode()` of one reference process per env, the batch
    their concatenation) — what one auto-reset rollout of nwin windows holds here, the unfinished tail of every stream
    discarded.  Actions are SAMPLED, not taped: the reference's own log-probs (float32) through the build's Philox
    inverse-CDF draw (oracle.sample_one, counters = env, the env's episode and step: what ic3_policy_step draws), so that the
    one-launch rollout — which samples inside the launch — plays the same episodes."""
    import oracle
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    a = rh.make_args(env_name, max_steps=T, seed=seed, **flags)
    env = rh.make_env(env_name, a)
    rh.finish_args(a, env)
    torch.manual_seed(seed)
    if model == 'mlp':        # the reference's IC / IRIC baselines (models.py:8-97; main.py:161-168 builds them like this)
        net = ref['models'].MLP(a, a.num_inputs)
    elif model == 'rnn':
        net = ref['models'].RNN(a, a.num_inputs)
    else:
        net = ref['comm'].CommNetMLP(a, a.num_inputs)
    if closed_form:
        sd = net.state_dict()
        cw = closed_form_weights({k: tuple(v.shape) for k, v in sd.items()})
        net.load_state_dict({k: torch.from_numpy(cw[k]) for k in sd})
    tr = ref['trainer'].Trainer(a, net, env)
    N, nh = a.nagents, len(a.naction_heads)
    cursor = {}
    trmod = ref['trainer']
    played = []

    def sampled_select(args, action_out):
        e, ep, t = cursor['e'], cursor['ep'], cursor['t']
        cursor['t'] += 1
        if env_name == 'traffic_junction':
            ref['rnd'].begin(cursor['st'], philox.DOMAIN_TJ_ADD, ep, t)
        act = np.zeros((nh, N), np.int64)
        for h in range(nh):
            lp = action_out[h].detach().numpy().reshape(N, -1).astype(np.float32)
            for n in range(N):
                act[h, n] = oracle.sample_one(lp[n], philox.x24(seed, 400 + e, philox.DOMAIN_SAMPLE, ep, t, h * N + n))
        played.append((e, ep, t, act.copy()))
        return torch.from_numpy(act).view(nh, 1, N, 1)
    trmod.select_action = sampled_select
    batch = []
    stats = dict(num_episodes=0)
    streams = [philox.Stream(seed, 400 + e) for e in range(nenv)]
    envs = [rh.make_env(env_name, a) for e in range(nenv)]
    lengths = []
    for e in range(nenv):
        tr.env = envs[e]
        slots, ep, lens = 0, 0, []
        while True:
            cursor.update(e=e, ep=ep, t=0, st=streams[e])
            ref['rnd'].begin(streams[e], philox.DOMAIN_PP_RESET, ep, 0)
            n0 = len(played)
            episode, stat = tr.get_episode(ep)
            if slots + len(episode) > nwin * T:          # does not end inside the batch: discarded
                del played[n0:]
                break
            slots += len(episode)
            lens.append(len(episode))
            ref['utils'].merge_stat(stat, stats)
            stats['num_episodes'] += 1
            batch += episode
            ep += 1
        lengths.append(lens)
    trmod.select_action = ref['action_utils'].select_action
    stats['num_steps'] = len(batch)
    batch = trmod.Transition(*zip(*batch))
    tr.optimizer.zero_grad()
    s = tr.compute_grad(batch)
    maxep = max(len(l) for l in lengths)
    ep_len = np.zeros((nenv, maxep), np.int32)
    for e, l in enumerate(lengths):
        ep_len[e, :len(l)] = l
    acts = np.full((nenv, nwin * T, nh, N), -1, np.int32)     # per env, slot by slot (diagnostics for a diverging draw)
    off = {}
    for e, ep, t, act in played:
        base = int(ep_len[e, :ep].sum())
        acts[e, base + t] = act
    out = dict(cfg=np.array([N, T, nenv, nwin, nh, seed], np.int32), flags=np.array(repr(sorted(flags.items()))),
               ep_len=ep_len, actions=acts, action_loss=s['action_loss'], value_loss=s['value_loss'],
               entropy=s.get('entropy', 0.0), num_steps=stats['num_steps'], num_episodes=stats['num_episodes'],
               reward=np.asarray(stats['reward'], np.float64), success=float(stats.get('success', 0)))
    if model:
        out['model'] = np.array(model)
    if closed_form:
        out['param_names'] = np.array(list(net.state_dict().keys()))
        out['param_shapes'] = np.array([repr(tuple(v.shape)) for v in net.state_dict().values()])
    else:
        for k, v in net.state_dict().items():
            out['w:' + k] = v.detach().numpy().copy()
    for k, p in net.named_parameters():
        g = np.zeros(0) if p.grad is None else p.grad.detach().numpy().copy()
        out['g:' + k] = g.astype(np.float32) if closed_form else g
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'steps', stats['num_steps'], 'episodes', stats['num_episodes'], 'lengths', lengths, 'losses', s)


def grad_baseline_main():
    """F5b for the non-communicating baselines (round-4 verdict: their native update was only checked against this repo's
    autograd): the reference's run_batch + compute_grad with models.MLP (IC), models.RNN with the tanh recurrence and with the
    LSTM cell (IRIC), hid 64."""
    grad_case('grad_pp_medium_ic_mlp', 'predator_prey', 20, 3, 2, 61, model='mlp', nagents=5, dim=10, vision=1, hid_size=64,
              recurrent=False, entr=0.01, value_coeff=0.01)
    grad_case('grad_pp_medium_iric_rnn', 'predator_prey', 20, 3, 1, 62, model='rnn', nagents=5, dim=10, vision=1, hid_size=64,
              recurrent=True, rnn_type='MLP', detach_gap=8, entr=0.01, value_coeff=0.01, normalize_rewards=True)
    grad_case('grad_tj_easy_iric_lstm', 'traffic_junction', 20, 3, 2, 63, model='rnn', nagents=5, dim=6, vision=1, hid_size=64,
              recurrent=True, rnn_type='LSTM', detach_gap=6, add_rate_min=0.3, add_rate_max=0.3, difficulty='easy', entr=0.01,
              value_coeff=0.01)


def grad_stream_main():
    """Collection-mode fixtures (round-4 verdict item 5): a tiny Predator-Prey grid on which sampled policies DO end episodes
    early (cuts inside the windows, a different number of episodes per env, detach points of the env's own step counter), and
    Traffic-Junction (episodes of exactly max_steps: the cuts fall on the window borders; alive masks and the gate head)."""
    grad_stream_case('gradstream_pp_tiny_ic3net', 'predator_prey', 12, 6, 3, 51, nagents=2, dim=3, vision=1, hid_size=16,
                     ic3net=True, recurrent=True, detach_gap=5, entr=0.01, value_coeff=0.01, mode='mixed')
    grad_stream_case('gradstream_pp_tiny_commnet', 'predator_prey', 10, 5, 2, 52, nagents=2, dim=3, vision=0, hid_size=16,
                     commnet=True, recurrent=True, detach_gap=4, normalize_rewards=True, mean_ratio=0.5, gamma=0.9,
                     mode='mixed')
    grad_stream_case('gradstream_tj_easy_ic3net', 'traffic_junction', 10, 4, 2, 53, nagents=5, dim=6, vision=0, hid_size=16,
                     ic3net=True, recurrent=True, detach_gap=4, add_rate_min=0.3, add_rate_max=0.3, difficulty='easy')


def grad_stream_h128_main():
    """Collection mode at the hidden size of the BASELINE configs (round-5 verdict item 2a): hid 128, so that the kernels the
    headline update runs — lstm_gates_bwd_kernel<128, 1, 1> with its per-row cuts, comm_bwd_kernel<128>, lstm_wgrad_kernel<128> —
    are compared with the reference on the device where the cuts fall INSIDE a window: a small Predator-Prey grid on which the
    sampled policy does end episodes early (16 envs, 3 windows of 20 slots, detach points of the env's own step counter every 7
    steps), and Traffic-Junction (alive masks, the gate head; episodes of exactly max_steps, the detach points every 3 steps
    fall mid-window).  Closed-form weights (the fixture stores no weights)."""
    grad_stream_case('gradstream_pp_small_h128', 'predator_prey', 20, 16, 3, 54, closed_form=True, nagents=3, dim=3, vision=1,
                     hid_size=128, ic3net=True, recurrent=True, detach_gap=7, entr=0.01, value_coeff=0.01, mode='mixed')
    grad_stream_case('gradstream_tj_easy_h128', 'traffic_junction', 10, 6, 2, 55, closed_form=True, nagents=5, dim=6, vision=1,
                     hid_size=128, ic3net=True, recurrent=True, detach_gap=3, add_rate_min=0.3, add_rate_max=0.3,
                     difficulty='easy', entr=0.01, value_coeff=0.01)


def grad_stream_families_main():
    """Collection mode for the other policy families (round-5 verdict item 6), hid 64 (a size their one-launch rollouts run at),
    closed-form weights: the NON-recurrent CommNet module with two communication passes, the gated non-recurrent module on
    Traffic-Junction, the IC baseline (models.MLP) and the IRIC baseline with the LSTM cell and with the tanh recurrence
    (models.RNN) — on the tiny
    Predator-Prey grid where sampled policies end episodes early, and Traffic-Junction (alive masks)."""
    grad_stream_case('gradstream_pp_tiny_commnet_mlp2', 'predator_prey', 12, 6, 3, 67, closed_form=True, nagents=2, dim=3, vision=1,
                     hid_size=64, commnet=True, recurrent=False, comm_passes=2, entr=0.01, value_coeff=0.01, mode='mixed')
    grad_stream_case('gradstream_tj_easy_ic3net_mlp', 'traffic_junction', 10, 4, 2, 57, closed_form=True, nagents=5, dim=6, vision=1,
                     hid_size=64, ic3net=True, recurrent=False, add_rate_min=0.3, add_rate_max=0.3, difficulty='easy', entr=0.01,
                     value_coeff=0.01)
    grad_stream_case('gradstream_pp_tiny_ic_mlp', 'predator_prey', 12, 6, 3, 58, closed_form=True, model='mlp', nagents=2, dim=3,
                     vision=1, hid_size=64, recurrent=False, entr=0.01, value_coeff=0.01, mode='mixed')
    grad_stream_case('gradstream_pp_tiny_iric_lstm', 'predator_prey', 12, 6, 3, 59, closed_form=True, model='rnn', nagents=2, dim=3,
                     vision=1, hid_size=64, recurrent=True, rnn_type='LSTM', detach_gap=5, entr=0.01, value_coeff=0.01,
                     mode='mixed')
    grad_stream_case('gradstream_pp_tiny_iric_rnn', 'predator_prey', 12, 6, 3, 64, closed_form=True, model='rnn', nagents=2, dim=3,
                     vision=1, hid_size=64, recurrent=True, rnn_type='MLP', detach_gap=5, entr=0.01, value_coeff=0.01, mode='mixed')
    grad_stream_case('gradstream_pp_tiny_ic3net_p2', 'predator_prey', 12, 6, 3, 60, closed_form=True, nagents=2, dim=3, vision=1,
                     hid_size=64, ic3net=True, recurrent=True, comm_passes=2, detach_gap=5, entr=0.01, value_coeff=0.01,
                     mode='mixed')


def trainer_main():
    grad_case('grad_pp_easy_ic3net', 'predator_prey', 20, 4, 2, 31, nagents=3, dim=5, vision=0, hid_size=16,
              ic3net=True, recurrent=True, detach_gap=10, entr=0.01, value_coeff=0.01)
    grad_case('grad_pp_medium_commnet_norm', 'predator_prey', 40, 3, 1, 32, nagents=5, dim=10, vision=1, hid_size=16,
              commnet=True, recurrent=True, detach_gap=10, normalize_rewards=True, mean_ratio=0.5, gamma=0.95)
    grad_case('grad_tj_easy_ic3net', 'traffic_junction', 20, 4, 2, 33, nagents=5, dim=6, vision=0, hid_size=16,
              ic3net=True, recurrent=True, detach_gap=10, add_rate_min=0.3, add_rate_max=0.3, difficulty='easy')
    grad_case('grad_tj_medium_perhead', 'traffic_junction', 40, 2, 1, 34, nagents=10, dim=14, vision=1, hid_size=16,
              ic3net=True, recurrent=True, detach_gap=10, add_rate_min=0.2, add_rate_max=0.2, difficulty='medium',
              advantages_per_action=True, entr=0.001)
    trainer_case('trainer_pp_easy', 'predator_prey', 20, 3, 2, 21, greedy=True, nagents=3, dim=5, vision=0,
                 hid_size=16, ic3net=True, recurrent=True, detach_gap=10)
    trainer_case('trainer_pp_medium', 'predator_prey', 40, 2, 2, 22, greedy=True, nagents=5, dim=10, vision=1,
                 hid_size=16, ic3net=True, recurrent=True, detach_gap=10)
    trainer_case('trainer_tj_easy', 'traffic_junction', 20, 3, 2, 23, nagents=5, dim=6, vision=0, hid_size=16,
                 ic3net=True, recurrent=True, detach_gap=10, add_rate_min=0.3, add_rate_max=0.3, difficulty='easy')
    trainer_case('trainer_tj_medium_commnet', 'traffic_junction', 40, 2, 2, 24, nagents=10, dim=14, vision=1,
                 hid_size=16, commnet=True, recurrent=True, detach_gap=10, add_rate_min=0.2, add_rate_max=0.2,
                 difficulty='medium')
    trainer_fullsize_main()


def grad_fullsize_main():
    """F5b at BASELINE shapes (round-3 verdict item 4): the reference's compute_grad for PP-hard (configs[1]: 10 agents,
    dim 20, vision 1, hid 128, 80 steps, detach_gap 10; env 0 is steered onto the prey and ends early) and TJ-hard
    (configs[3]: 20 agents, dim 18, hid 128, 80 steps) on closed-form weights."""
    grad_case('grad_pp_hard_ic3net', 'predator_prey', 80, 3, 1, 41, closed_form=True, nagents=10, dim=20, vision=1,
              hid_size=128, ic3net=True, recurrent=True, detach_gap=10, entr=0.01, value_coeff=0.01)
    grad_case('grad_tj_hard_ic3net', 'traffic_junction', 80, 2, 1, 42, closed_form=True, nagents=20, dim=18, vision=1,
              hid_size=128, ic3net=True, recurrent=True, detach_gap=10, add_rate_min=0.05, add_rate_max=0.05,
              difficulty='hard', entr=0.01, value_coeff=0.01)


def grad_scaled_main():
    """F5b at the hidden size / agent count of config 5 (round-5 verdict item 4): 32 agents, hid 256, vision 2, closed-form
    weights — on a 12 x 12 grid and 8 steps, so that encoder.weight's gradient (hid x obs_dim) stays a small fixture; the native
    update at hid 256 is the recomputing backward (ic3_lstm_gates_backward<256>, the kernel config 5 runs)."""
    grad_case('grad_pp_scaled_h256_ic3net', 'predator_prey', 8, 2, 1, 45, closed_form=True, nagents=32, dim=12, vision=2,
              hid_size=256, ic3net=True, recurrent=True, detach_gap=3, entr=0.01, value_coeff=0.01)


def grad_nonrec_main():
    """F5b for the NON-recurrent module (comm.py:127-129,220-224): the reference's compute_grad with
    recurrent = False — CommNet with two communication passes, and the gated (IC3Net-style) module with shared weights."""
    grad_case('grad_pp_medium_commnet_mlp2', 'predator_prey', 20, 3, 1, 43, closed_form=True, nagents=5, dim=10, vision=1,
              hid_size=64, commnet=True, recurrent=False, comm_passes=2, entr=0.01, value_coeff=0.01)
    grad_case('grad_tj_easy_ic3net_mlp2share', 'traffic_junction', 20, 3, 1, 44, closed_form=True, nagents=5, dim=6, vision=1,
              hid_size=64, ic3net=True, recurrent=False, comm_passes=2, share_weights=True, add_rate_min=0.3, add_rate_max=0.3,
              difficulty='easy', entr=0.01, value_coeff=0.01)


def trainer_fullsize_main():
    """F5 at BASELINE shapes: PP-hard (configs[1]) and TJ-hard (configs[3]), IC3Net recurrent hid 128, 80 steps."""
    trainer_case('trainer_pp_hard', 'predator_prey', 80, 2, 1, 25, greedy=True, nagents=10, dim=20, vision=1,
                 hid_size=128, ic3net=True, recurrent=True, detach_gap=10)
    trainer_case('trainer_tj_hard', 'traffic_junction', 80, 2, 1, 26, nagents=20, dim=18, vision=1, hid_size=128,
                 ic3net=True, recurrent=True, detach_gap=10, add_rate_min=0.05, add_rate_max=0.05, difficulty='hard')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'fullsize':
        fullsize_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'multipass':
        multipass_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'nonrec':
        nonrec_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'trainer':
        trainer_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_nonrec':
        grad_nonrec_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_scaled':
        grad_scaled_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_fullsize':
        grad_fullsize_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_baseline':
        grad_baseline_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_stream':
        grad_stream_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_stream_families':
        grad_stream_families_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'grad_stream_h128':
        grad_stream_h128_main()
    elif len(sys.argv) > 1 and sys.argv[1] == 'trainer_fullsize':
        trainer_fullsize_main()
    else:
        main()
