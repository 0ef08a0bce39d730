#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on its headline config, on N MI355X of one node.

metric   env-steps/sec counted as agents x envs x steps (BASELINE.json) of the rollout hot path
         Trainer.step_episode = {CommNetMLP forward -> select_action -> env.step (step + obs assembly kernels)},
         episodes reset inside the timed region every max_steps steps.
workload configs[1] "Predator-Prey hard": 10 agents, dim 20, vision 1, max_steps 80, IC3Net recurrent hid 128,
         8192 parallel envs per GPU (weak scaling: every rank owns 8192 envs, global env ids rank*8192 + e).
A "step" = one lock-step iteration of the hot loop over all envs of the rank.

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without a launcher: bench.py starts the N
                                                                     ranks itself through torch.distributed.run)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W               (the driver's form: one rank per GPU over RCCL)
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak (f32 in / f32 acc), same guide
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix peak (no sparsity), same guide


WORKLOADS = {
    # name: (env_name, dict of flags)  — BASELINE.json configs
    "pp_hard": ("predator_prey", dict(nagents=10, dim=20, vision=1, max_steps=80, hid_size=128, ic3net=True,
                                      recurrent=True, detach_gap=10, mode='mixed')),
    "pp_easy": ("predator_prey", dict(nagents=3, dim=5, vision=0, max_steps=20, hid_size=128, ic3net=True,
                                      recurrent=True, detach_gap=10, mode='mixed')),
    "tj_medium": ("traffic_junction", dict(nagents=10, dim=14, vision=1, max_steps=40, hid_size=128, commnet=True,
                                           recurrent=True, detach_gap=10, difficulty='medium', add_rate_min=0.05,
                                           add_rate_max=0.05)),
    "tj_hard": ("traffic_junction", dict(nagents=20, dim=18, vision=1, max_steps=80, hid_size=128, ic3net=True,
                                         recurrent=True, detach_gap=10, difficulty='hard', add_rate_min=0.05,
                                         add_rate_max=0.05)),
    # the NON-recurrent CommNet module (comm.py:127-129,220-224) on the TJ-medium env, two communication
    # passes — one launch per step through ic3_commnet_step
    "tj_medium_commnet_mlp": ("traffic_junction", dict(nagents=10, dim=14, vision=1, max_steps=40, hid_size=128, commnet=True,
                                                       recurrent=False, comm_passes=2, difficulty='medium', add_rate_min=0.05,
                                                       add_rate_max=0.05)),
    # the reference's non-communicating baselines (models.py:8-97) on the PP-hard env — IC = models.MLP,
    # IRIC = models.RNN with the LSTM cell — one launch per step through their kernel stand-ins (ic3net_amd/models.py)
    "pp_hard_ic": ("predator_prey", dict(nagents=10, dim=20, vision=1, max_steps=80, hid_size=128, recurrent=False,
                                         baseline='mlp', detach_gap=10, mode='mixed')),
    "pp_hard_iric": ("predator_prey", dict(nagents=10, dim=20, vision=1, max_steps=80, hid_size=128, recurrent=True,
                                           rnn_type='LSTM', baseline='rnn', detach_gap=10, mode='mixed')),
    "pp_hard_iric_tanh": ("predator_prey", dict(nagents=10, dim=20, vision=1, max_steps=80, hid_size=128, recurrent=True,
                                                rnn_type='MLP', baseline='rnn', detach_gap=10, mode='mixed')),
    "pp_scaled": ("predator_prey", dict(nagents=32, dim=40, vision=2, max_steps=80, hid_size=256, ic3net=True,
                                        recurrent=True, detach_gap=10, mode='mixed')),
}


def make_args(env_name, flags, nenvs, seed, env_id_offset, device):
    """main.py:22-155 argument handling for the flags the hot path reads."""
    a = argparse.Namespace(
        batch_size=500, hid_size=64, recurrent=False, seed=seed, lrate=0.001, env_name=env_name, max_steps=20,
        display=False, commnet=False, ic3net=False, nagents=1, comm_mode='avg', comm_passes=1, comm_mask_zero=False,
        mean_ratio=1.0, rnn_type='MLP', detach_gap=10000, comm_init='uniform', hard_attn=False, comm_action_one=False,
        share_weights=False, nenvs=nenvs, env_id_offset=env_id_offset, device=device, store_states=False,
        hip_graph=False)
    if env_name == 'predator_prey':
        a.__dict__.update(nenemies=1, dim=5, vision=2, moving_prey=False, no_stay=False, mode='mixed', enemy_comm=False)
    else:
        a.__dict__.update(dim=5, vision=1, add_rate_min=0.05, add_rate_max=0.2, curr_start=0, curr_end=0,
                          difficulty='easy', vocab_type='bool')
    a.__dict__.update(flags)
    if a.ic3net:                          # main.py:115-123
        a.commnet = 1
        a.hard_attn = 1
        a.mean_ratio = 0
        if a.env_name == 'traffic_junction':
            a.comm_action_one = True
    a.nfriendly = a.nagents               # main.py:125
    return a


def build_trainer(workload, nenvs, seed, env_id_offset, device, **overrides):
    import torch
    from ic3net_amd import data
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    from ic3net_amd.trainer import Trainer
    env_name, flags = WORKLOADS[workload]
    flags = dict(flags, **overrides)
    a = make_args(env_name, flags, nenvs, seed, env_id_offset, device)
    env = data.init(env_name, a, False)
    a.num_actions = [env.num_actions]     # main.py:134-152
    a.dim_actions = env.dim_actions
    a.num_inputs = env.observation_dim
    if a.hard_attn and a.commnet:
        a.num_actions = [*a.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    torch.manual_seed(seed)               # default PyTorch init, random weights (no checkpoints offline)
    if getattr(a, 'baseline', None):      # main.py:161-168: MLP / RNN instead of CommNetMLP
        from ic3net_amd import models
        a.continuous = False
        net = (models.RNN if a.baseline == 'rnn' else models.MLP)(a, a.num_inputs)
    else:
        net = CommNetMLP(a, a.num_inputs)
    net = net.to(torch.device('cuda', device)).float()
    return Trainer(a, net, env), a


# --------------------------------------------------------------------------------------------------
# CPU baseline leg: the oracle ("port" of the reference's per-env Python/numpy loop), timed on the host
# cores of this box on a bounded sample of the same workload.  One env object per step call, batch-1
# fp64 policy calls — the reference's cost structure — in `procs` forked workers (README: nprocesses 16).
# --------------------------------------------------------------------------------------------------
# Probe numbers of the ACTUAL reference (fp64 torch, CPU), measured in the build container with
# tests/golden/bench_reference.py and BASELINE.md §2's harness — NOT on the GPU box (the reference cannot travel);
# reported next to the same-box legs, labelled with their host.
REFERENCE_PROBE = {
    "host": "build container, 8 x Intel Xeon @ 2.10 GHz, torch 2.10 CPU fp64, OMP_NUM_THREADS=1 (BASELINE.md §2)",
    "unit": "agent-steps/s",
    "pp_hard": {"rollout_1proc": 5580, "train_batch_1proc": 2872, "train_batch_8proc": 14693, "train_batch_16proc": 14311},
    "pp_easy": {"rollout_1proc": 3732, "train_batch_1proc": 1981, "train_batch_16proc": 11656},
    "tj_medium": {"rollout_1proc": 7570, "train_batch_1proc": 4520, "train_batch_16proc": 23512},
    "tj_hard": {"rollout_1proc": 9320, "train_batch_1proc": 6041, "train_batch_16proc": 18269},
}


def _cpu_worker(job):
    workload, env_ids, episodes, seed, budget_s = job[:5]
    shaped = len(job) > 5 and job[5]          # leg (ii): the reference-shaped numpy env (dense one-hot copy per step)
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    import oracle
    from oracle import policy_ref, philox
    env_name, f = WORKLOADS[workload]
    N, H, T = f['nagents'], f['hid_size'], f['max_steps']
    rs = np.random.RandomState(seed)
    t0 = time.perf_counter()
    steps = 0
    params = None
    for gid in env_ids:
        if time.perf_counter() - t0 > budget_s:
            break
        if env_name == 'predator_prey' and shaped:
            from oracle.pp_numpy import PPNumpyEnv
            env = PPNumpyEnv(N, f['dim'], f['vision'], f['mode'], seed=seed, env_gid=gid)
        elif env_name == 'predator_prey':
            env = oracle.PPOracle(N, f['dim'], f['vision'], f['mode'], seed=seed, env_gid=gid)
        else:
            env = oracle.TJOracle(N, f['dim'], f['vision'], f['difficulty'], add_rate_min=f['add_rate_min'],
                                  add_rate_max=f['add_rate_max'], seed=seed, env_gid=gid)
        heads = [5 if env_name == 'predator_prey' else 2] + ([2] if f.get('ic3net') else [])
        if params is None:
            k = 1.0 / np.sqrt(H)
            shp = {'encoder.weight': (H, env.obs_dim), 'encoder.bias': (H,), 'f_module.weight_ih': (4 * H, H),
                   'f_module.weight_hh': (4 * H, H), 'f_module.bias_ih': (4 * H,), 'f_module.bias_hh': (4 * H,),
                   'C_modules.0.weight': (H, H), 'C_modules.0.bias': (H,), 'value_head.weight': (1, H),
                   'value_head.bias': (1,)}
            for i, A in enumerate(heads):
                shp['heads.%d.weight' % i] = (A, H)
                shp['heads.%d.bias' % i] = (A,)
            params = {n: rs.uniform(-k, k, size=s) for n, s in shp.items()}
        for ep in range(episodes):
            obs = env.reset() if env_name == 'predator_prey' else env.reset(ep)
            hc = (np.zeros((N, H)), np.zeros((N, H)))
            alive, ca = None, np.zeros(N)
            for t in range(T):
                logp, value, hc = policy_ref.forward(params, obs[None].astype(np.float64), hc, alive, ca,
                                                     recurrent=True, hard_attn=bool(f.get('ic3net')),
                                                     nheads=len(heads))
                acts = []
                for hd, lp in enumerate(logp):
                    acts.append([oracle.sample_one(lp[0, n].astype(np.float32),
                                                   philox.x24(seed, gid, philox.DOMAIN_SAMPLE, ep, t, hd * N + n))
                                 for n in range(N)])
                obs, rew, done = env.step(np.array(acts[0]))
                if env_name == 'traffic_junction':
                    alive = env.alive.astype(np.float64)
                    ca = np.ones(N)
                else:
                    ca = np.array(acts[-1], np.float64)
                steps += 1
                if done:
                    break
    return steps, time.perf_counter() - t0


def _cpu_leg(workload, procs, envs_per_proc, episodes, seed, budget_s, shaped):
    import multiprocessing as mp
    jobs = [(workload, list(range(p * envs_per_proc, (p + 1) * envs_per_proc)), episodes, seed, budget_s, shaped)
            for p in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    return sum(r[0] for r in res), wall


def cpu_baseline(workload, envs_per_proc=256, episodes=1, seed=0, budget_s=10.0):
    """The three legs of SURVEY §8(d), on a bounded sample (every worker plays whole episodes on fresh envs until
    `budget_s` of CPU time is spent):
      (i)   "port": the C oracle env + batch-1 fp64 numpy policy in 16 forked processes  -> value / unit / cores
      (ii)  "reference_shaped": the same loop with the reference-shaped numpy env (one dense one-hot grid copy per step,
            oracle/pp_numpy.py) — Predator-Prey workloads
      (iii) "reference_probe": numbers of the actual reference from the build container, labelled with their host."""
    sys.path.insert(0, ROOT)
    import oracle
    oracle.build()
    procs = max(1, min(16, os.cpu_count() or 1))        # the reference's nprocesses=16 (README.md:46)
    N = WORKLOADS[workload][1]['nagents']
    env_steps, wall = _cpu_leg(workload, procs, envs_per_proc, episodes, seed, budget_s, False)
    out = {"value": round(N * env_steps / wall, 1), "unit": "agent-steps/s", "cores": procs, "kind": "port",
           "sample": "%d procs x whole %s episodes for %.0f s each, batch-1 fp64 numpy policy + C oracle env: "
                     "%d env-steps in %.1f s wall" % (procs, workload, budget_s, env_steps, wall)}
    if WORKLOADS[workload][0] == 'predator_prey':
        es2, wall2 = _cpu_leg(workload, procs, envs_per_proc, episodes, seed, budget_s, True)
        out["reference_shaped"] = {
            "value": round(N * es2 / wall2, 1), "unit": "agent-steps/s", "cores": procs,
            "sample": "%d procs, reference-shaped numpy env (dense one-hot grid copy per step) + batch-1 fp64 numpy "
                      "policy: %d env-steps in %.1f s wall" % (procs, es2, wall2)}
    probe = REFERENCE_PROBE.get(workload)
    if probe:
        out["reference_probe"] = dict(probe, host=REFERENCE_PROBE["host"], unit=REFERENCE_PROBE["unit"])
    return out


def mfma_roofline(a, nenvs, step_ms, gate_split=False):
    """The MFMA-bound kernel of the step: policy_step_kernel.  Algorithmic flops per launch = the two dense layers of
    comm.py (C: H x H, LSTMCell: 2H x 4H) plus the heads, 2 flops per multiply-add, for E*N rows: `achieved` / `frac` are
    these fp32-EQUIVALENT flops against the fp32 matrix peak whatever the gate product runs on.  With the split gate product
    (nine bf16 x bf16 products per fp32 product) `bf16_issued` reports what the bf16 matrix cores were actually asked for:
    9 x the gate product's flops against the dense bf16 peak."""
    R, H = nenvs * a.nagents, a.hid_size
    OT = sum(int(x) for x in a.naction_heads) + 1
    rec = bool(getattr(a, 'recurrent', True)) and getattr(a, 'rnn_type', 'LSTM') == 'LSTM'
    gate = 2.0 * R * (2 * H * 4 * H) if rec else 0.0
    # non-recurrent module (comm.py:220-224): per communication pass one [comm | h] . [C_i | F_i]^T product (2H x H)
    flops = gate + 2.0 * R * (H * H + H * OT) if rec else 2.0 * R * (int(a.comm_passes) * 2 * H * H + H * OT)
    if getattr(a, 'baseline', None):      # models.py:23-34 / 75-84: affine2 (H x H) or the LSTM cell, + heads — no C layer
        flops = gate + 2.0 * R * H * OT if rec else 2.0 * R * (H * H + H * OT)
    avg = sum(step_ms) / len(step_ms)
    tf = flops / (avg * 1e-3) / 1e12
    out = {"kernel": "policy_step_kernel" if rec else "commnet_forward_kernel<H, env> (ic3_commnet_step)", "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
           "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "flops_per_launch": flops,
           "flops_counted": "fp32-equivalent (2 per multiply-add of comm.py's dense layers)",
           "avg_launch_ms": round(avg, 4), "launches": len(step_ms),
           "hbm_bytes_per_launch_algorithmic": R * ((4 * H if rec else 0) + OT + 2 * len(a.naction_heads) + 1) * 4}
    if gate_split and rec:
        btf = 9.0 * gate / (avg * 1e-3) / 1e12
        out["bf16_issued"] = {"flops_per_launch": 9.0 * gate, "achieved": round(btf, 1), "peak": MFMA_BF16_PEAK_TF,
                              "unit": "TFLOP/s", "frac": round(btf / MFMA_BF16_PEAK_TF, 4),
                              "note": "gate product only: 9 bf16 x bf16 products per fp32 product on "
                                      "v_mfma_f32_32x32x16_bf16; the C product and the heads stay on the fp32 instruction"}
    return out


def train_mode(o, trainer, a, raw_env, rank, world, use_dist, backend, red_dev, barrier, cpu):
    """--mode train: a step = one Trainer.train_batch (trainer.py:245-256: run_batch over max_steps lock-step steps of every
    env, compute_grad, RMSprop; with more than one rank the gradient all-reduce of multi_processing.py:86-97 over RCCL).  W
    untimed updates, then EXACTLY K updates between barrier + synchronize; `value` = agent-steps of the K updates / time.  The
    roofline object is the dominant kernel of the backward, lstm_gates_bwd_kernel<H, 1, 1> (cell derivative from the recorded
    gates + input gradient), event-timed live inside the timed updates (HIP events recorded on the launch stream around every
    gate launch by ic3_bptt_backward itself)."""
    import torch
    import torch.distributed as dist
    T, N, H, E = a.max_steps, a.nagents, a.hid_size, o.nenvs
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0.0, value_coeff=0.01, advantages_per_action=False,
                      batch_size=E * T, hip_graph=False, auto_reset=bool(o.auto_reset), gate_split=bool(o.gate_split))
    native = trainer._native_update()
    gc.disable()
    for u in range(max(1, o.warmup)):
        trainer.train_batch(u)
    raw_env.gate_timer = []
    torch.cuda.reset_peak_memory_stats()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0.0
    for u in range(o.steps):
        st = trainer.train_batch(o.warmup + u)
        steps += st['num_steps']              # (summed over the ranks by train_batch's stat all-reduce)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    gate_ms_chain = [s_.elapsed_time(e_) for s_, e_, _t in raw_env.gate_timer]
    # In the timed updates the backward's launches run as TWO concurrent chains of envs (ic3_bptt.two_chains): the events there
    # bracket the first chain's gate launch while the second chain's kernels share the GPU.  The kernel ALONE: two more updates
    # behind the timed region with the backward as one chain, every gate launch of them event-timed.
    two = bool(getattr(a, 'bptt_two_chains', True))
    a.bptt_two_chains = False
    raw_env.gate_timer = []
    for u in range(2):
        trainer.train_batch(o.warmup + o.steps + u)
    torch.cuda.synchronize()
    gate_ms = [s_.elapsed_time(e_) for s_, e_, _t in raw_env.gate_timer]
    raw_env.gate_timer = None
    a.bptt_two_chains = two
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return
    OT = sum(int(x) for x in a.naction_heads) + 1
    R = E * N
    roofline = mf = None
    if gate_ms:
        # per agent row: the recorded gates in, dgates out (in place), c_prev, dL/dh, dL/dc in, dL/dc_prev out, [d inp | d h] out,
        # the heads' dL/dlogits row in
        nbytes = R * (4 * H + 4 * H + 3 * H + H + 2 * H + OT) * 4
        avg = sum(gate_ms) / len(gate_ms)
        gbs = nbytes / (avg * 1e-3) / 1e9
        roofline = {"kernel": "lstm_gates_bwd_kernel<%d, 1, 1> (cell derivative from the recorded gates, the heads' share folded in, "
                              "input gradient dgates . [W_ih | W_hh] in the same launch)" % H, "bound": "hbm",
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": None, "bytes_per_launch": nbytes, "avg_launch_ms": round(avg, 4), "launches": len(gate_ms),
                    "launch_ms_min": round(min(gate_ms), 4), "launch_ms_max": round(max(gate_ms), 4),
                    "timed": "HIP events recorded on the launch stream around every gate launch of two updates behind the timed "
                             "region, the backward as ONE chain of launches (the kernel alone on the GPU)"}
        if two and gate_ms_chain:
            from ic3net_amd import ops as _ops
            E1 = _ops.first_chain_envs(E, N)
            if E1 < E:
                avg1 = sum(gate_ms_chain) / len(gate_ms_chain)
                roofline["in_the_timed_updates"] = {
                    "what": "the first chain's gate launch (envs [0, %d)) while the second chain's launches share the GPU" % E1,
                    "bytes_per_launch": nbytes * E1 // E, "avg_launch_ms": round(avg1, 4), "launches": len(gate_ms_chain),
                    "GBps": round(nbytes * E1 / E / (avg1 * 1e-3) / 1e9, 1)}
        flops = 2.0 * R * 4 * H * 2 * H
        mf = {"kernel": "lstm_gates_bwd_kernel (its input-gradient product)", "bound": "mfma",
              "achieved": round(flops / (avg * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
              "frac": round(flops / (avg * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4), "flops_per_launch": flops,
              "flops_counted": "fp32-equivalent; issued as nine exact bf16 x bf16 products per fp32 product",
              "bf16_issued_frac": round(9.0 * flops / (avg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}
    if cpu is not None:
        cpu = dict(cpu, note="the same-box legs time the ROLLOUT of the port (no CPU update half is built: the reference's "
                             "update is PyTorch autograd); the reference's own train_batch numbers are under reference_probe")
    out = {
        "metric": "env-steps/sec (agents x envs x steps), Trainer.train_batch: rollout + backward through time + RMSprop",
        "value": round(N * steps / dt, 1), "unit": "agent-steps/s", "n_gpus": world, "steps": o.steps, "warmup": o.warmup,
        "ms_per_step": round(dt / o.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": o.workload, "mode": "train: a step = one train_batch update of %d envs x %d steps" % (E, T),
                   "envs_per_gpu": E, "agents": N, "env_steps_per_update": int(steps / o.steps) if o.steps else 0,
                   "parallelism": "env-shard x%d, gradient all-reduce at update time" % world,
                   "update": ("no-grad one-launch rollout recording (h, c), gates, inp; explicit backward through time: "
                              "ic3_bptt_backward (2 hand-written launches per step and chain of envs, two chains on two streams, "
                              "one host call per window) + ic3_env_encode_backward_window + ic3_lstm_weight_grad per window — "
                              "no library GEMM" if native else "autograd"),
                   "auto_reset": bool(o.auto_reset)},
        "roofline": roofline, "roofline_mfma": mf, "cpu_baseline": cpu,
        "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "collectives": backend,
    }
    print(json.dumps(out))


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=None, help='timed steps (default 160; --mode train: 12 updates)')
    p.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default 16; --mode train: 2 updates)')
    p.add_argument('--mode', default='rollout', choices=['rollout', 'train'],
                   help="rollout (BASELINE.json's metric, the default): a step = one lock-step iteration of the hot loop; train: a "
                        "step = one Trainer.train_batch update (rollout of max_steps steps + backward through time + RMSprop, "
                        "trainer.py:245-256) — the second headline (round-5 verdict), same one-line JSON")
    p.add_argument('--episode-graph', type=int, default=1,
                   help='1: behind the timed region, also time the Trainer\'s DEFAULT execution mode — Trainer.get_episode with one '
                        'hipGraph per episode — and report it as value_episode_graph (what ships; the eager event-timed loop stays '
                        'the headline because the roofline needs its launches timed)')
    p.add_argument('--workload', default='pp_hard', choices=sorted(WORKLOADS))
    p.add_argument('--nenvs', type=int, default=8192, help='environments per GPU')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--dispatch-events', type=int, default=1,
                   help='1: the HIP events that time the step launch are stamped by the dispatch itself '
                        '(hipExtLaunchKernel); 0: recorded into the stream before / after it')
    p.add_argument('--time-every', type=int, default=0,
                   help='event-time every N-th step launch of the timed region (stamping both HIP events costs ~10 us per '
                        'launch, measured: eager launches without events run 3 % faster); 0 = min(4, max(1, steps // 8))')
    p.add_argument('--graph', type=int, default=int(os.environ.get('IC3_BENCH_GRAPH', '1')),
                   help='replay the per-step launch sequence as hipGraphs (Trainer args.hip_graph)')
    p.add_argument('--no-dense-obs', action='store_true',
                   help='diagnostic: skip obs assembly (sparse encoder consumes env state directly); NOT the headline config')
    p.add_argument('--overlap-obs', type=int, default=int(os.environ.get('IC3_BENCH_OVERLAP_OBS', '0')),
                   help='assemble the dense observation on a second stream, overlapped with the next step (graph mode)')
    p.add_argument('--mega', type=int, default=int(os.environ.get('IC3_BENCH_MEGA', '1')),
                   help='policy forward + action draws + env.step as ONE launch (ic3_policy_step); 0 = the launch chain')
    p.add_argument('--time-kernels', type=int, default=int(os.environ.get('IC3_BENCH_TIME_KERNELS', '1')),
                   help='bracket the policy+step launch and the obs-assembly launch with HIP events in the timed region '
                        '(roofline numbers); the two launches are then issued eagerly instead of as graph replays')
    p.add_argument('--auto-reset', type=int, default=0,
                   help='1: finished envs restart inside the step launch (collection mode of trainer.py:227-242); every '
                        'slot of the timed region is then a live transition')
    p.add_argument('--fused-obs', type=int, default=int(os.environ.get('IC3_BENCH_FUSED_OBS', '1')),
                   help='next_state rows stored by the policy+step launch itself (0: separate obs-assembly launch)')
    p.add_argument('--gate-split', type=int, default=1,
                   help='1 (default): the gate product of ic3_policy_step with every fp32 operand split exactly into three '
                        'bf16 terms, all nine cross products on the bf16 matrix cores, fp32 accumulation (exact products: '
                        'fp32-class arithmetic, DESIGN.md); 0: the fp32 matrix instruction')
    p.add_argument('--incremental-obs', type=int, default=0,
                   help='EXPERIMENT (labelled in the output, never the headline): ic3_policy_step maintains the obs rows '
                        'incrementally (clears what the previous step painted, paints the new entries) instead of '
                        'zero-filling them every step; the rows are bit-identical, the HBM traffic is not')
    p.add_argument('--rccl', type=int, default=int(os.environ.get('IC3_BENCH_RCCL', '0')),
                   help='1: bring up the RCCL process group even for one rank (world_size 1) so that the timing barrier '
                        'and the MAX / SUM reductions of the N > 1 path run on device tensors over RCCL')
    p.add_argument('--tune-gemm', type=int, default=int(os.environ.get('IC3_BENCH_TUNE_GEMM', '1')),
                   help='let PyTorch TunableOp pick the fastest hipBLASLt/rocBLAS solution for the two policy GEMMs '
                        'during the eager warm-up episode (seconds; selections are kept in memory)')
    o = p.parse_args()
    if o.steps is None:
        o.steps = 12 if o.mode == 'train' else 160
    if o.warmup is None:
        o.warmup = 2 if o.mode == 'train' else 16

    if o.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the
        # reference's one worker per process: multi_processing.py:41-72) and hand over — rank 0 of the children prints
        # the ONE JSON line.  Exactly what the driver's torch.distributed.run command line does.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(o.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if 'IC3_BENCH_DEVICE' in os.environ:      # test hook: several ranks on one GPU (exercises the world>1 control flow)
        local_rank = int(os.environ['IC3_BENCH_DEVICE'])
    assert world == o.gpus, "WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run --nproc-per-node == --gpus, " \
                            "or without a launcher (bench.py then starts the ranks itself)" % (world, o.gpus)

    cpu = None
    if rank == 0 and world == 1 and not o.no_cpu_baseline:
        cpu = cpu_baseline(o.workload)            # before CUDA is initialised (fork-safe)

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)

    if o.mega:
        o.tune_gemm = 0        # no library GEMM on the one-launch path (the chain fallback uses the default heuristics)
    if o.tune_gemm:
        try:
            import torch.cuda.tunable as tunable
            if hasattr(tunable, 'write_file_on_exit'):          # absent in some torch builds
                tunable.write_file_on_exit(False)
            tunable.enable(True)
            tunable.tuning_enable(True)
            tunable.set_filename(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ic3_tunableop_%d.csv' % os.getpid()))
        except Exception as exc:
            sys.stderr.write("bench.py: TunableOp unavailable (%r); using the default GEMM heuristics\n" % (exc,))
            o.tune_gemm = 0
            try:                                     # leave nothing half-enabled
                torch.cuda.tunable.tuning_enable(False)
                torch.cuda.tunable.enable(False)
            except Exception:
                pass
    trainer, a = build_trainer(o.workload, o.nenvs, o.seed, rank * o.nenvs, local_rank)
    a.hip_graph = bool(o.graph)
    a.dense_obs = not o.no_dense_obs
    a.overlap_obs = bool(o.overlap_obs)
    a.mega_policy = bool(o.mega)
    a.fused_obs = bool(o.fused_obs)
    a.auto_reset = bool(o.auto_reset)
    a.incremental_obs = bool(o.incremental_obs)
    a.gate_split = bool(o.gate_split)
    T = a.max_steps
    raw_env = trainer.env.env
    live_done = [0.0]                         # live env-steps of the episodes that ENDED so far (stat['num_steps'])

    def run(nsteps, t_in_ep):
        for _ in range(nsteps):
            if t_in_ep == 0:
                trainer.begin_episode(0)
            trainer.step_episode(t_in_ep)
            t_in_ep += 1
            if t_in_ep == T:
                live_done[0] += trainer.end_episode()[1]['num_steps']   # stats reduced on device, one host read per episode
                t_in_ep = 0
        return t_in_ep

    # Host-side set-up that idles the GPU goes FIRST (an idle MI355X clocks down and needs several ms of work to come
    # back: a 10 ms timed region right behind an idle gap would be measured on the ramp): the full CPython GC pass
    # (35-80 ms over torch + numpy + the CPU-baseline imports; survivors are frozen so that no collection lands in the
    # timed region), then RCCL bring-up (only the timing barrier / max-reduce use it — no collective on the rollout
    # path).  The untimed eager + capture episodes below then double as the warm-up of the clocks, and the W warm-up
    # steps and the timed region follow them without a gap.
    gc.collect()
    gc.freeze()
    backend = None
    use_dist = world > 1 or bool(o.rccl)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))   # (only when no launcher set one)
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        if 'IC3_BENCH_DEVICE' in os.environ:       # test hook (several ranks on one GPU): RCCL refuses duplicate devices
            dist.init_process_group(backend='gloo', rank=rank, world_size=world)
            dist.barrier()
            backend = 'gloo'
        else:
            try:
                # the rank's GPU is named explicitly (LOCAL_RANK -> device): nothing is inferred from the environment
                dist.init_process_group(backend='nccl', rank=rank, world_size=world,
                                        device_id=torch.device('cuda', local_rank))
                dist.barrier(device_ids=[local_rank])
                backend = 'nccl'
            except Exception as exc:                   # timing barrier only: gloo is an acceptable stand-in
                sys.stderr.write("bench.py: RCCL bring-up failed (%r); using gloo for the timing barrier\n" % (exc,))
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group(backend='gloo', rank=rank, world_size=world)
                dist.barrier()
                backend = 'gloo'
    red_dev = 'cuda' if backend == 'nccl' else 'cpu'

    def barrier():
        if use_dist:
            if backend == 'nccl':
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
    if o.mode == 'train':
        train_mode(o, trainer, a, raw_env, rank, world, use_dist, backend, red_dev, barrier, cpu)
        if use_dist:
            dist.destroy_process_group()
        return
    raw_env.obs_timer = []                    # event-time the obs launch from the start (graphs are captured in this mode)
    if o.tune_gemm:                           # untimed: every GEMM shape is met (and tuned) in one eager episode
        saved_graph, a.hip_graph = a.hip_graph, False
        run(T, 0)
        a.hip_graph = saved_graph
        torch.cuda.tunable.tuning_enable(False)   # keep the selections, stop tuning (never tune inside a capture)
    if o.graph:                               # untimed: one eager episode (warm-up) + one capture episode
        try:
            run(2 * T, 0)
        except Exception as exc:              # capture trouble on this box: measure the eager path instead of dying
            sys.stderr.write("bench.py: hipGraph capture failed (%r); falling back to eager launches\n" % (exc,))
            torch.cuda.synchronize()
            trainer, a = build_trainer(o.workload, o.nenvs, o.seed, rank * o.nenvs, local_rank)
            a.hip_graph, a.dense_obs, o.graph = False, not o.no_dense_obs, 0
            a.mega_policy = bool(o.mega)
            a.fused_obs = bool(o.fused_obs)
            raw_env = trainer.env.env
            raw_env.obs_timer = []
    else:
        run(T, 0)                             # no graphs: one untimed eager episode (first-use set-up, clocks)
    mega_live = bool(o.mega) and (getattr(trainer.policy_net, 'mega_steps', 0) > 0 or
                                  getattr(trainer.policy_net, 'commnet_steps', 0) > 0)   # the one-launch path is in use
    if o.time_kernels and mega_live:
        raw_env.step_timer = []               # the launch of every step is event-timed and issued eagerly
        raw_env.dispatch_events = bool(o.dispatch_events)   # events stamped by the dispatch, not recorded around it
        raw_env.step_timer_every = o.time_every if o.time_every > 0 else min(4, max(1, o.steps // 8))
    gc.disable()                              # (like timeit: no collector pause in the warm-up + timed steps)
    t_in_ep = run(o.warmup, 0)                # W untimed warm-up steps, in the measured configuration

    def timed_region(t_in_ep):
        """EXACTLY o.steps steps between barrier + synchronize on both sides; returns this rank's wall time, the
        event-timed launch durations inside it and the live env-steps it simulated."""
        raw_env.obs_timer = []
        if raw_env.step_timer is not None:
            raw_env.step_timer = []
        live0 = live_done[0] + raw_env.device_stats().live_env_steps  # (synchronises) finished episodes + the running one
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_in_ep = run(o.steps, t_in_ep)
        host_dt = time.perf_counter() - t0    # host-side enqueue time (diagnostic: host- vs GPU-bound)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        obs_ms = [s_.elapsed_time(e_) for s_, e_ in raw_env.obs_timer]
        step_all = [(s_.elapsed_time(e_), t_) for s_, e_, t_ in (raw_env.step_timer or [])]
        live = live_done[0] + raw_env.device_stats().live_env_steps - live0
        return t_in_ep, dt, host_dt, obs_ms, step_all, live

    # One 6 ms sample (the driver's 20 steps) can be hit by a clock ramp or a host hiccup.  The region is therefore
    # checked against the device's own clock: the event-timed launches inside it must account for the wall time
    # (GPU-bound loop: wall = launches + a few us of gaps per step).  A region whose two clocks disagree by more than
    # 10 % is measured again (at most 3 attempts, each EXACTLY o.steps steps); the attempt count is reported.
    attempts = []
    for attempt in range(3):
        t_in_ep, dt, host_dt, obs_ms, step_all, live_steps = timed_region(t_in_ep)
        # (every N-th step launch is event-timed: the launches of the region = their mean x the steps)
        launch_sum = (sum(ms for ms, _ in step_all) / len(step_all) * o.steps if step_all else 0.0) + sum(obs_ms)
        # (tolerance: 10 % of the wall time, or the 15 us per step that the gap between two dependent launches can cost —
        #  the larger; a 70 us launch such as PP-easy's is otherwise flagged for its launch gaps alone)
        consistent = (not step_all) or abs(dt * 1e3 - launch_sum) <= max(0.10 * dt * 1e3, 0.015 * o.steps)
        attempts.append(dict(ms_per_step=round(dt / o.steps * 1e3, 4), launches_ms=round(launch_sum / o.steps, 4),
                             consistent=bool(consistent)))
        all_ok = consistent
        if use_dist:                          # every rank repeats or none does
            flag = torch.tensor([0.0 if consistent else 1.0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            all_ok = float(flag.item()) == 0.0
        if all_ok:
            break
    gc.enable()
    # the store stream alone, same run: the stand-alone obs-assembly kernel on the same rows (rank 0, default configuration)
    store_ref_ms = []
    if rank == 0 and step_all and not o.no_dense_obs and not o.incremental_obs:
        raw_env.obs_timer = []
        for _ in range(12):
            raw_env.observe_timed()
        torch.cuda.synchronize()
        store_ref_ms = [s_.elapsed_time(e_) for s_, e_ in raw_env.obs_timer][2:]
    # What SHIPS: Trainer.get_episode in its default execution mode — the episode's max_steps launches as ONE hipGraph, the
    # per-episode statistics read included — timed over whole episodes (>= the steps of the region above), every rank.
    ep_graph = None
    if o.episode_graph and mega_live and not (o.overlap_obs or o.incremental_obs):
        raw_env.obs_timer = None
        raw_env.step_timer = None
        a.hip_graph = True
        try:
            n_ep = max(2, (o.steps + T - 1) // T)
            trainer.get_episode(0)
            trainer.get_episode(0)            # (the capture, when the episode above was the first in this mode)
            replayed = 'episode' in trainer._graphs
            barrier()
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            g_live = 0.0
            for _ in range(n_ep):
                g_live += trainer.get_episode(0)[1]['num_steps']
            torch.cuda.synchronize()
            barrier()
            g_dt = time.perf_counter() - tg0
            if use_dist:
                gt = torch.tensor([g_dt], dtype=torch.float64, device=red_dev)
                dist.all_reduce(gt, op=dist.ReduceOp.MAX)
                gl = torch.tensor([g_live], dtype=torch.float64, device=red_dev)
                dist.all_reduce(gl, op=dist.ReduceOp.SUM)
                g_dt, g_live = float(gt.item()), float(gl.item())
            ep_graph = dict(value=round(a.nagents * g_live / g_dt, 1), unit="agent-steps/s", episodes=n_ep, steps=n_ep * T,
                            ms_per_step=round(g_dt / (n_ep * T) * 1e3, 4), one_graph_per_episode=bool(replayed),
                            loop="Trainer.get_episode (args.hip_graph, the default of ic3net_amd.main): reset + %d step launches "
                                 "replayed as one hipGraph + the episode's masks / statistics, per episode" % T)
        except Exception as exc:              # never lose the headline line over the second field
            sys.stderr.write("bench.py: the episode-graph loop failed (%r); value_episode_graph omitted\n" % (exc,))
            torch.cuda.synchronize()
    rank_ms = [dt / o.steps * 1e3]
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)                          # per-rank times: a straggler is visible next to the MAX
        rank_ms = [float(g.item()) / o.steps * 1e3 for g in gathered]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    step_ms = [ms for ms, t_ in step_all if t_ > 0]            # t = 0 adds the h, c resets
    if o.auto_reset:                          # every slot is a real transition (the step counters restart in-launch)
        live_steps = float(o.nenvs * o.steps)
    if use_dist:
        lt = torch.tensor([live_steps], dtype=torch.float64, device=red_dev)
        dist.all_reduce(lt, op=dist.ReduceOp.SUM)
        live_steps = float(lt.item())
    if rank == 0:
        N = a.nagents
        E_total = o.nenvs * world
        # agent-steps actually simulated: envs whose episode ended early are frozen until the lock-step reset and do
        # not count (a random-init PP-hard policy never ends early: live_frac = 1)
        value = N * live_steps / dt
        live_frac = live_steps / float(E_total * o.steps)
        obs_bytes = o.nenvs * N * raw_env.obs_dim * 4          # algorithmic bytes of one obs-assembly launch
        fused_obs = bool(step_ms) and not obs_ms and not o.no_dense_obs    # next_state rows stored by policy_step_kernel
        # The HBM-bound kernel of the step and ITS launches inside the timed region.  Nothing is reported from launches
        # outside the per-step path (round 3's graph-mode line fell back to the two reset launches of pp_obs_kernel), and a
        # fraction is only formed over bytes the timed launches really wrote.
        state_bytes = mfma_roofline(a, o.nenvs, step_ms or [1.0])["hbm_bytes_per_launch_algorithmic"]
        per_step_obs = len(obs_ms) >= max(1, o.steps // 2)
        roof_note = None
        if fused_obs and o.incremental_obs:
            hbm_kernel, hbm_bytes, hbm_ms = "policy_step_kernel", None, step_ms
            roof_note = "EXPERIMENT --incremental-obs: the launch does not rewrite the rows, so no fraction is formed over them"
        elif fused_obs:
            lstm_pol = bool(a.recurrent) and getattr(a, 'rnn_type', 'LSTM') == 'LSTM'      # (the tanh-recurrence RNN runs ic3_commnet_step)
            hbm_kernel = ("policy_step_kernel" if lstm_pol else "commnet_forward_kernel<H, env>") + \
                " (policy + draws + env.step + obs assembly in one launch)"
            hbm_bytes, hbm_ms = obs_bytes + state_bytes, step_ms
        elif step_ms and o.no_dense_obs:
            hbm_kernel, hbm_bytes, hbm_ms = "policy_step_kernel (no obs rows: diagnostic)", state_bytes, step_ms
        elif per_step_obs:
            hbm_kernel = "pp_obs_kernel" if a.env_name == 'predator_prey' else "tj_obs_kernel"
            hbm_bytes, hbm_ms = obs_bytes, obs_ms
        else:
            hbm_kernel, hbm_bytes, hbm_ms = None, None, []
            roof_note = ("the step's launches were not event-timed (--time-kernels 0: hipGraph replays, or plain eager "
                         "launches): run with --time-kernels 1 for the roofline of the step's launch")
        avg_ms = sum(hbm_ms) / max(len(hbm_ms), 1)
        achieved = hbm_bytes / (avg_ms * 1e-3) / 1e9 if (hbm_ms and hbm_bytes) else None
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'obs_traffic.json')
        if os.path.exists(tf) and hbm_kernel and not o.incremental_obs:
            try:
                tab = json.load(open(tf))
                key = o.workload + ('_fused' if fused_obs else '')
                traffic = tab.get(key) if o.nenvs == tab.get('_nenvs', 8192) else None   # measured at that size
            except Exception:
                traffic = None
        roofline = None if hbm_kernel is None else {
            "kernel": hbm_kernel, "bound": "hbm", "achieved": round(achieved, 1) if achieved is not None else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved is not None else None,
            # (HBM traffic by PMC counters is NOT measured by this command: a bench run cannot collect them.  What separate
            #  rocprofv3 --pmc passes of this command recorded is quoted under `from_profiles`, named as what it is; null here.)
            "traffic": None,
            "from_profiles": {"traffic_bytes_per_launch": traffic, "source": "profiles/obs_traffic.json: rocprofv3 --pmc "
                              "WRITE_SIZE / FETCH_SIZE passes of this command in an earlier call, gfx950 read correction "
                              "applied"} if traffic is not None else None,
            "bytes_per_launch": hbm_bytes, "obs_bytes_per_launch": obs_bytes if not o.no_dense_obs else 0,
            "avg_launch_ms": round(avg_ms, 4), "launches": len(hbm_ms), "note": roof_note}
        if roofline is not None and fused_obs and achieved is not None and not o.incremental_obs and store_ref_ms:
            # context next to `frac` (which stays algorithmic bytes / the 8 TB/s spec peak), MEASURED IN THIS RUN: the stand-alone
            # obs-assembly kernel (ic3_env_observe: nothing but the same obs rows, zeros + non-zero entries) event-timed right
            # behind the timed region — the store stream this chip takes when nothing else runs beside it.
            R_ = o.nenvs * N
            written = obs_bytes + R_ * ((2 * a.hid_size if lstm_pol else (a.hid_size if a.recurrent else 0)) + sum(int(x) for x in a.naction_heads) + 1
                                        + 2 * len(a.naction_heads) + 1) * 4
            ref_ms = sum(store_ref_ms) / len(store_ref_ms)
            ref_gbs = obs_bytes / (ref_ms * 1e-3) / 1e9
            roofline["store_stream_reference"] = {
                "kernel": ("pp_obs_kernel" if a.env_name == 'predator_prey' else "tj_obs_fill/vec4_kernel") +
                          " (ic3_env_observe alone, the same obs rows)",
                "bytes_per_launch": obs_bytes, "launches": len(store_ref_ms), "avg_launch_ms": round(ref_ms, 4),
                "min_launch_ms": round(min(store_ref_ms), 4), "GBps": round(ref_gbs, 1),
                "written_bytes_per_step_launch": written,
                "step_launch_write_rate_GBps": round(written / (avg_ms * 1e-3) / 1e9, 1),
                "step_launch_write_rate_over_reference": round(written / (avg_ms * 1e-3) / 1e9 / ref_gbs, 4),
                "measured": "in this run, HIP events, behind the timed region"}
        out = {
            "metric": "env-steps/sec (agents x envs x steps), rollout hot path",
            "value": round(value, 1), "unit": "agent-steps/s", "n_gpus": world, "steps": o.steps, "warmup": o.warmup,
            "ms_per_step": round(dt / o.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Predator-Prey hard: 10 agents, dim 20, vision 1, max_steps 80, IC3Net recurrent "
                                   "hid 128, %d envs per GPU" % o.nenvs if o.workload == 'pp_hard' else o.workload,
                       "envs_per_gpu": o.nenvs, "agents": N, "obs_dim": raw_env.obs_dim, "parallelism": "env-shard x%d" % world,
                       "launch": ("eager, event-timed: ONE launch per step" if fused_obs else
                                  "eager, event-timed: policy+step launch, obs launch" if step_ms else
                                  "hipGraph replay" if o.graph else "eager"),
                       "dense_obs": not o.no_dense_obs, "overlap_obs": bool(o.overlap_obs),
                       "policy": ("one launch per step (%s)" % ("ic3_policy_step" if (a.recurrent and getattr(a, 'rnn_type', 'LSTM') == 'LSTM') else "ic3_commnet_step"))
                       if mega_live else "launch chain",
                       "auto_reset": bool(o.auto_reset),
                       "obs_rows": ("EXPERIMENT: maintained incrementally (not rewritten every step) - not the headline "
                                    "configuration" if o.incremental_obs else
                                    "rewritten every step"),
                       "gemm": ("hand-written MFMA; gate product [inp|h].[W_ih|W_hh]^T: every fp32 operand split exactly into 3 "
                                "bf16 terms, all 9 cross products on v_mfma_f32_32x32x16_bf16 (each product exact in fp32), fp32 "
                                "accumulation - fp32-class arithmetic (error vs fp64 = the fp32 instruction's, "
                                "tests/test_gate_split_gpu.py; --gate-split 0 selects v_mfma_f32_32x32x2_f32); C product and "
                                "heads: fp32 MFMA" if (mega_live and o.gate_split and a.recurrent) else
                                "hand-written fp32 MFMA (v_mfma_f32_32x32x2_f32)" if mega_live else
                                "TunableOp-selected" if o.tune_gemm else "default heuristics")},
            "live_frac": round(live_frac, 6),
            "value_episode_graph": ep_graph,
            "roofline": roofline, "roofline_note": roof_note if roofline is None else None,
            "cpu_baseline": cpu,
            "roofline_mfma": mfma_roofline(a, o.nenvs, step_ms, bool(mega_live and o.gate_split)) if step_ms else None,
            "host_enqueue_ms_per_step": round(host_dt / o.steps * 1e3, 4),
            "ms_per_step_ranks": [round(x, 4) for x in rank_ms],
            "collectives": backend,
            "timing": {"attempts": attempts, "consistent": attempts[-1]["consistent"],
                       "launch_ms_min": round(min(step_ms), 4) if step_ms else None,
                       "launch_ms_median": round(sorted(step_ms)[len(step_ms) // 2], 4) if step_ms else None,
                       "launch_ms_max": round(max(step_ms), 4) if step_ms else None,
                       "event_timed_launches": len(step_ms),
                       "event_timed_every": getattr(raw_env, 'step_timer_every', 1) if step_ms else None},
        }
        if not attempts[-1]["consistent"]:
            out["timing_inconsistent"] = True     # no attempt had wall clock and device clock agree: not a valid headline
            sys.stderr.write("bench.py: wall clock and event-timed launches disagree by more than max(10 %%, 15 us per step) in all %d "
                             "attempts: %r\n" % (len(attempts), attempts))
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
