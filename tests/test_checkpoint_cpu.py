"""CPU: checkpoint / log layout compatibility with /root/reference/main.py:260-272."""
import sys
import types
from collections import namedtuple

import numpy as np
import torch

from policy_util import PolicyCase


def _policy(pc):
    from ic3net_amd.comm import CommNetMLP
    return CommNetMLP(pc.args(), pc.obs_dim)


class _Opt(object):
    def __init__(self, net):
        self.optimizer = torch.optim.RMSprop(net.parameters(), lr=0.001, alpha=0.97, eps=1e-6)

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, s):
        self.optimizer.load_state_dict(s)


def test_roundtrip(tmp_path):
    from ic3net_amd import checkpoint
    pc = PolicyCase("policy_ic3net_small")
    net, tr, log = _policy(pc), None, checkpoint.new_log()
    tr = _Opt(net)
    assert list(log) == ['epoch', 'reward', 'enemy_reward', 'success', 'steps_taken', 'add_rate', 'comm_action',
                         'enemy_comm', 'value_loss', 'action_loss', 'entropy']
    log['epoch'].data.extend([1, 2])
    log['reward'].data.extend([np.array([-1.0, -2.0]), np.array([-0.5, -1.0])])
    checkpoint.save(str(tmp_path / "ck.pt"), net, log, tr)
    net2, log2 = _policy(pc), checkpoint.new_log()
    tr2 = _Opt(net2)
    d = checkpoint.load(str(tmp_path / "ck.pt"), net2, log2, tr2)
    assert set(d) == {'policy_net', 'log', 'trainer'}
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    assert log2['epoch'].data == [1, 2] and log2['reward'].divide_by == 'num_episodes'


def test_loads_a_reference_written_checkpoint(tmp_path):
    """A file exactly as the reference writes it: float64 state_dict with the reference's key set (from the
    policy fixtures), `log` holding `utils.LogField` namedtuples, RMSprop state in float64."""
    from ic3net_amd import checkpoint
    pc = PolicyCase("policy_ic3net_small")
    ref_utils = types.ModuleType('utils')                      # stands in for the reference's utils module
    ref_utils.LogField = namedtuple('LogField', ('data', 'plot', 'x_axis', 'divide_by'))
    ref_utils.LogField.__module__ = 'utils'
    sys.modules['utils'] = ref_utils
    try:
        sd = {k: torch.from_numpy(np.asarray(v)).double() for k, v in pc.params.items()}
        ref_net = _policy(pc).double()
        ref_net.load_state_dict(sd)
        opt = torch.optim.RMSprop(ref_net.parameters(), lr=0.001, alpha=0.97, eps=1e-6)
        (ref_net.encoder.weight.sum() + ref_net.value_head.weight.sum()).backward()
        opt.step()
        log = {'epoch': ref_utils.LogField([1], False, None, None),
               'reward': ref_utils.LogField([np.array([-1.5])], True, 'epoch', 'num_episodes')}
        torch.save({'policy_net': ref_net.state_dict(), 'log': log, 'trainer': opt.state_dict()},
                   str(tmp_path / "ref.pt"))
    finally:
        del sys.modules['utils']
    net, mylog = _policy(pc), checkpoint.new_log()
    tr = _Opt(net)
    checkpoint.load(str(tmp_path / "ref.pt"), net, mylog, tr)
    assert net.encoder.weight.dtype == torch.float32
    np.testing.assert_allclose(net.encoder.weight.detach().numpy(), ref_net.encoder.weight.detach().numpy(), atol=1e-6)
    assert mylog['epoch'].data == [1] and type(mylog['reward']).__module__ == 'ic3net_amd.utils'
    st = tr.optimizer.state_dict()['state']
    assert st and all(v['square_avg'].dtype == torch.float32 for v in st.values())
