"""ic3net_amd — MI355X-native batched rollout engine for IC3Net's data-parallel hot path.

Host-side mirror of the reference's interfaces for that path (same names / argument meaning / error
behaviour as /root/reference: data.init, PredatorPreyEnv, TrafficJunctionEnv, GymWrapper, CommNetMLP,
select_action, translate_action, Trainer.get_episode/run_batch) over the C ABI of
ic3net_amd/csrc/libic3rollout.so (include/ic3_rollout.h).  There is no CPU fallback: every env / op
call fails loudly if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
