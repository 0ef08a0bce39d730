"""data.py mirror (/root/reference/data.py:6-36): env factory, same names."""
from .env_wrappers import GymWrapper
from . import envs


def init(env_name, args, final_init=True):
    if env_name == 'predator_prey':            # data.py:16-21
        env = envs.PredatorPreyEnv()
    elif env_name == 'traffic_junction':       # data.py:22-27
        env = envs.TrafficJunctionEnv()
    elif env_name in ('levers', 'number_pairs', 'starcraft'):
        raise NotImplementedError("%s is not shipped with / in scope of the hot path (SURVEY §2 rows 13, 17)" % env_name)
    else:
        raise RuntimeError("wrong env name")
    # args.display: the reference calls env.init_curses() here (data.py:18-19,24-25); rendering is plain text in
    # this engine (env.render()), so there is nothing to initialise
    env.multi_agent_init(args)
    return GymWrapper(env)
