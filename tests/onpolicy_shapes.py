"""Restatement (for tests) of how the runner derives shapes from spaces by class NAME:
onpolicy/utils/util.py:32-52 get_shape_from_obs_space / get_shape_from_act_space."""


def get_shape_from_obs_space(obs_space):
    name = obs_space.__class__.__name__
    if name == "Box":
        return obs_space.shape
    if name == "list":
        return obs_space
    raise NotImplementedError


def get_shape_from_act_space(act_space):
    name = act_space.__class__.__name__
    if name == "Discrete":
        return 1
    if name == "MultiDiscrete":
        return act_space.shape
    if name in ("Box", "MultiBinary"):
        return act_space.shape[0]
    return act_space[0].shape[0] + 1
