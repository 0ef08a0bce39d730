"""ic3_env_encode_backward (the update half's gradient of comm.py:51,119's nn.Linear, trainer.py:128-225) against the
dense fp64 product obs^T @ g, on states reached by random play; plus the autograd wiring (ops.env_encode) and the
snapshot semantics (backward of an earlier step after the env has moved on)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("pp", dict(N=10, dim=20, v=1, H=128, E=96)), ("pp", dict(N=3, dim=5, v=0, H=64, E=33)),
         ("pp", dict(N=32, dim=40, v=2, H=256, E=5)), ("pp", dict(N=4, dim=7, v=1, H=32, E=700)),
         ("pp", dict(N=5, dim=8, v=1, H=32, E=64, enemy_comm=True)),
         ("pp", dict(N=4, dim=70, v=1, H=32, E=20)),       # 4900 grid cells: the per-env form (P does not fit in LDS)
         ("tj", dict(N=10, dim=14, v=1, diff="medium", H=128, E=64)), ("tj", dict(N=20, dim=18, v=0, diff="hard", H=128, E=32)),
         ("tj", dict(N=5, dim=6, v=1, diff="easy", H=32, E=17)), ("tj", dict(N=5, dim=6, v=1, diff="easy", H=32, E=1100)),
         ("tj", dict(N=10, dim=14, v=1, diff="medium", H=64, E=48, vocab_type="scalar"))]


def build(kind, c, seed=1):
    from test_env_parity_gpu import make_pp, make_tj
    if kind == "pp":
        env = make_pp(c['N'], c['dim'], c['v'], "mixed" if not c.get('enemy_comm') else "cooperative", c['E'], seed=seed,
                      **({'enemy_comm': True} if c.get('enemy_comm') else {}))
        env.reset()
        return env, 5
    env = make_tj(c['N'], c['dim'], c['v'], c['diff'], c['E'], seed=seed, add_rate_min=0.4, add_rate_max=0.4,
                  **({'vocab_type': c['vocab_type']} if c.get('vocab_type') else {}))
    env.reset(0)
    return env, 2


def play(env, nact, steps, gen):
    E, N = env.nenvs, env.nagents_env
    for _ in range(steps):
        env.step(torch.randint(0, nact, (E, N), device='cuda', dtype=torch.int32, generator=gen))


@pytest.mark.parametrize("kind,c", CASES)
def test_encode_backward_equals_dense(kind, c):
    env, nact = build(kind, c)
    gen = torch.Generator(device='cuda').manual_seed(3)
    H, R = c['H'], env.nenvs * env.nagents_env
    for steps in (0, 6, 7):
        play(env, nact, steps, gen)
        obs = env.observe().reshape(R, -1).double()
        g = torch.randn(R, H, device='cuda', generator=gen)
        dwt, db = env.encode_backward(g)
        ref = obs.t() @ g.double()
        scale = max(1.0, float(ref.abs().max()))
        assert float((dwt.double() - ref).abs().max()) <= 2e-6 * scale
        torch.testing.assert_close(db.double(), g.double().sum(0), atol=2e-6 * max(1.0, R ** 0.5), rtol=0)
        # strided grad_out (a column slice of a wider buffer) and no bias
        wide = torch.randn(R, 2 * H, device='cuda', generator=gen)
        dwt2, none = env.encode_backward(wide[:, H:], want_bias=False)
        assert none is None
        ref2 = obs.t() @ wide[:, H:].double()
        assert float((dwt2.double() - ref2).abs().max()) <= 2e-6 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("kind,c", CASES)
def test_encode_backward_accumulated_over_states_equals_the_sum(kind, c):
    """ic3_env_encode_backward_accumulate / _finish: the gradient over several states (snapshots of different steps) with one
    expansion at the end equals the sum of the per-state dense products; the configuration whose first stage has no
    partial-sums form (the 4900-cell grid) reports it and the caller keeps the per-state calls."""
    env, nact = build(kind, c, seed=2)
    gen = torch.Generator(device='cuda').manual_seed(5)
    H, R = c['H'], env.nenvs * env.nagents_env
    ref = None
    refb = torch.zeros(H, dtype=torch.float64, device='cuda')
    ok = None
    for k, steps in enumerate((0, 3, 4)):
        play(env, nact, steps, gen)
        snap = env.snapshot()
        obs = env.observe().reshape(R, -1).double()
        wide = torch.randn(R, 2 * H, device='cuda', generator=gen)
        g = wide[:, :H]                                            # a strided column slice, as bptt hands d inp
        play(env, nact, 1, gen)                                    # the live state moves on: the snapshot is what counts
        ok = env.encode_backward_accumulate(g, snap, first=(k == 0))
        if not ok:
            break
        r = obs.t() @ g.double()
        ref = r if ref is None else ref + r
        refb += g.double().sum(0)
    if kind == "pp" and c['dim'] == 70:
        assert ok is False
        return
    assert ok is True
    dwt, db = env.encode_backward_finish(H)
    scale = max(1.0, float(ref.abs().max()))
    assert float((dwt.double() - ref).abs().max()) <= 4e-6 * scale
    torch.testing.assert_close(db.double(), refb, atol=4e-6 * max(1.0, (3 * R) ** 0.5), rtol=0)


@pytest.mark.parametrize("kind,c", CASES)
def test_encode_backward_over_a_window_of_states_in_one_launch(kind, c):
    """ic3_env_encode_backward_window / _window_finish: stage 1 over T recorded states in ONE launch on the matrix cores (one-hot x
    gradient products, the gradient split exactly into three bf16 terms) = the sum of the per-state dense fp64 products; rows read
    out of a (T, R, 2H) ring as ic3_bptt_backward leaves them; two windows (the second adds)."""
    env, nact = build(kind, c, seed=3)
    gen = torch.Generator(device='cuda').manual_seed(7)
    H, R, T = c['H'], env.nenvs * env.nagents_env, 4
    assert env.encode_window_work(H) is not None
    ref = None
    refb = torch.zeros(H, dtype=torch.float64, device='cuda')
    for win in range(2):
        snaps = torch.empty((T, env.dims.state_words), dtype=torch.int32, device='cuda')
        ring = torch.randn(T, R, 2 * H, device='cuda', generator=gen)
        for t in range(T):
            play(env, nact, 2 + t, gen)
            snaps[t].copy_(env.snapshot())
            obs = env.observe().reshape(R, -1).double()
            r = obs.t() @ ring[t, :, :H].double()
            ref = r if ref is None else ref + r
            refb += ring[t, :, :H].double().sum(0)
        play(env, nact, 1, gen)
        env.encode_backward_window(ring, snaps, H, first=(win == 0))
    dwt, db = env.encode_backward_window_finish(H)
    scale = max(1.0, float(ref.abs().max()))
    assert float((dwt.double() - ref).abs().max()) <= 4e-6 * scale
    torch.testing.assert_close(db.double(), refb, atol=4e-6 * max(1.0, (2 * T * R) ** 0.5), rtol=0)


def test_snapshot_is_the_state_of_the_forward():
    env, nact = build("pp", CASES[0][1])
    gen = torch.Generator(device='cuda').manual_seed(5)
    play(env, nact, 4, gen)
    R, H = env.nenvs * env.nagents_env, 128
    obs_then = env.observe().reshape(R, -1).double().clone()
    snap = env.snapshot()
    play(env, nact, 5, gen)                                   # the env moves on
    g = torch.randn(R, H, device='cuda', generator=gen)
    dwt, _ = env.encode_backward(g, snap)
    ref = obs_then.t() @ g.double()
    assert float((dwt.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    now = env.observe().reshape(R, -1).double().t() @ g.double()
    assert float((dwt.double() - now).abs().max()) > 1e-3      # and it is not the current state's gradient


@pytest.mark.parametrize("kind,c", [CASES[0], CASES[5]])
def test_env_encode_autograd_matches_linear(kind, c):
    """Two-step chain through ops.env_encode: parameter gradients equal those of nn.Linear on the cloned observations."""
    from ic3net_amd import ops
    env, nact = build(kind, c)
    gen = torch.Generator(device='cuda').manual_seed(7)
    torch.manual_seed(0)
    lin = torch.nn.Linear(env.obs_dim, c['H']).cuda()
    ref = torch.nn.Linear(env.obs_dim, c['H']).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    loss, loss_ref = 0, 0
    for step in range(3):
        play(env, nact, 2, gen)
        mix = torch.randn(env.nenvs, env.nagents_env, c['H'], device='cuda', generator=gen)
        out = ops.env_encode(env, lin.weight, lin.bias)
        out_ref = ref(env.observe().double().clone())
        torch.testing.assert_close(out.double(), out_ref, atol=2e-6, rtol=0)
        loss = loss + (torch.tanh(out) * mix).sum()
        loss_ref = loss_ref + (torch.tanh(out_ref) * mix.double()).sum()
    loss.backward()
    loss_ref.backward()
    for p, q in zip(lin.parameters(), ref.parameters()):
        assert p.grad.is_contiguous()
        scale = max(1.0, float(q.grad.abs().max()))
        assert float((p.grad.double() - q.grad).abs().max()) <= 5e-6 * scale
