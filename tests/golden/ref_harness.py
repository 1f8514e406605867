"""Harness that imports the reference (read-only, /root/reference) to GENERATE golden vectors.

Runs ONLY in the build container (the reference never travels to the GPU box).  Nothing here is
imported by the product path or by the tests: tests read the .npz files this produces.

What it does:
  * installs inert import-time stubs for modules the reference imports but never executes on
    this path (gym, pyglet, jax, hj_reachability, cvxpy, casadi);
  * builds `MultiAgentGraphEnv` for the July tube scenario exactly the way the scenario's own
    `__main__` block does (nav_metered_one_goal_graph_rotate_tube_july.py:1877-1889), because the
    shipped `GraphMPEEnv` factory needs a callback this scenario lacks (MPE_env.py:79);
  * replaces `np.random.uniform` by an arithmetically identical wrapper that also LOGS the
    underlying [0,1) samples, so the engine's reset / goal-reach heading draws can be replayed
    from a "tape" (legacy RandomState.uniform == low + (high-low)*random_sample(), checked in
    `selfcheck_uniform_patch`).
"""
import argparse
import os
import sys
import types

import numpy as np

REF = os.environ.get("GMPE_REFERENCE", "/root/reference")


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Space(object):
        def __init__(self, *a, **k):
            pass

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete(_Space):
        def __init__(self, n):
            self.n = n

    class Tuple(_Space):
        def __init__(self, spaces):
            self.spaces = spaces

    class Env(object):
        def close(self):
            pass

    spaces = mod("gym.spaces", Box=Box, Discrete=Discrete, Tuple=Tuple, prng=types.SimpleNamespace())
    reg = mod("gym.envs.registration", register=lambda *a, **k: None)
    envs = mod("gym.envs", registration=reg)
    mod("gym", Env=Env, Space=_Space, spaces=spaces, envs=envs)
    mod("pyglet.image")
    mod("pyglet", image=sys.modules["pyglet.image"])
    jnp = mod("jax.numpy")
    mod("jax", numpy=jnp)
    mod("hj_reachability")

    class ControlAndDisturbanceAffineDynamics(object):
        pass

    mod("hj_reachability_utils.common", get_hj_grid_from_meta_data=lambda *a, **k: None,
        ControlAndDisturbanceAffineDynamics=ControlAndDisturbanceAffineDynamics)
    mod("hj_reachability_utils", common=sys.modules["hj_reachability_utils.common"])
    mod("cvxpy")
    mod("casadi")
    if REF not in sys.path:
        sys.path.insert(0, REF)


class UniformTape(object):
    """Logs the [0,1) samples behind every np.random.uniform call (legacy global RandomState)."""

    def __init__(self):
        self.samples = []
        self._orig = None

    def __enter__(self):
        self._orig = np.random.uniform

        def uniform(low=0.0, high=1.0, size=None):
            u = np.random.random_sample(size)
            self.samples.extend(np.atleast_1d(u).ravel().tolist())
            return low + (high - low) * u

        np.random.uniform = uniform
        return self

    def __exit__(self, *exc):
        np.random.uniform = self._orig


def selfcheck_uniform_patch():
    np.random.seed(7)
    a = [np.random.uniform(-0.3, 1.7), np.random.uniform(-4, 4, 2), np.random.uniform(0, 2 * np.pi)]
    np.random.seed(7)
    with UniformTape():
        b = [np.random.uniform(-0.3, 1.7), np.random.uniform(-4, 4, 2), np.random.uniform(0, 2 * np.pi)]
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y)), "uniform patch is not bit-identical"


def july_args(num_agents=3, world_size=4.0, episode_length=25, scenario_name="nav_metered_one_goal_graph_rotate_tube_july", **over):
    """Working arg set of SURVEY.md §8(c)."""
    d = dict(scenario_name=scenario_name, dynamics_type="air_taxi",
             world_size=world_size, num_agents=num_agents, num_landmarks=num_agents,
             num_scripted_agents=0, num_obstacles=0, num_walls=0, collaborative=False, max_speed=2,
             collision_rew=5, formation_rew=1, goal_rew=5, use_dones=False,
             episode_length=episode_length, num_env_steps=10000, n_rollout_threads=1,
             render_episodes=None, fair_wt=1, fair_rew=1, formation_type="point", total_actions=5,
             zeroshift=5, graph_feat_type="relative", discrete_action=True, use_safety_filter=False)
    d.update(over)
    return argparse.Namespace(**d)


def make_july_env(args):
    """Same construction as the scenario's own __main__ (…_july.py:1877-1889)."""
    install_stubs()
    from multiagent.custom_scenarios import load
    from multiagent.environment import MultiAgentGraphEnv
    scenario = load(args.scenario_name + ".py").Scenario()
    world = scenario.make_world(args)
    env = MultiAgentGraphEnv(world=world, reset_callback=scenario.reset_world,
                             reward_callback=scenario.reward,
                             observation_callback=scenario.observation,
                             graph_observation_callback=scenario.graph_observation,
                             info_callback=scenario.info_callback,
                             done_callback=scenario.done,
                             id_callback=scenario.get_id,
                             update_graph=scenario.update_graph,
                             agent_reached_goal_callback=getattr(scenario, "get_agent_reached_goal", None),
                             scenario_name=args.scenario_name,
                             discrete_action=args.discrete_action,
                             dynamics_type=args.dynamics_type)
    return env, scenario, world


def snapshot(env, scenario, world):
    """Internal state the engine carries."""
    ag = world.agents
    tp = world.tube_params
    return dict(
        x=np.array([a.state.p_pos[0] for a in ag]), y=np.array([a.state.p_pos[1] for a in ag]),
        theta=np.array([a.state.theta for a in ag]), speed=np.array([a.state.speed for a in ag]),
        status=np.array([bool(a.status) for a in ag]),
        prev_phase=np.array([getattr(a, "previous_phase", 0) for a in ag], dtype=np.int32),
        phase_reached=np.array(scenario.phase_reached, dtype=np.int32),
        goal_tracker=np.array(scenario.goal_tracker, dtype=np.int32),
        cooldown=np.array(scenario.entry_reward_cooldown, dtype=np.int32),
        prev_proj=np.array(getattr(scenario, "prev_proj", np.zeros(len(ag))), dtype=np.float64),
        p_dist=np.array([a.state.p_dist for a in ag]), time=np.array([a.state.time for a in ag]),
        current_step=np.int32(env.current_step),
        tube_angle=np.float64(tp["angle"]), entrance=np.array(tp["entrance"]), exit=np.array(tp["exit"]),
        tube_e=np.array(tp["e"], dtype=np.float64), tube_n=np.array(tp["n"], dtype=np.float64),
        tube_L=np.float64(tp["L"]), half_w=np.float64(tp["half_width"]), width=np.float64(tp["width"]),
        landmarks=np.array([l.state.p_pos for l in world.landmarks]),
    )
