"""formation_gym - MI355X-native drop-in for jc-bao/gym-formation's package API
(reference formation_gym/__init__.py:6-17).

    import formation_gym
    env = formation_gym.make_env('formation_hd_env', benchmark=False, num_agents=27,
                                 num_envs=4096, device='cuda:0')
    obs = env.reset()                                   # [B, N, 6N] float32 on the GPU
    obs, rew, done, info = env.step(actions)            # actions [B, N, 2]

With num_envs == 1 (the default) `reset()` / `step(list_of_arrays)` return the
reference's own Python-list shapes.  The hot path runs in hand-written HIP
kernels (csrc/formation_hip.hip) behind a C ABI (include/formation_hip.h); there
is no CPU fallback.
"""
import importlib.util
import os.path as osp

from .environment import MultiAgentEnv
from .scenario import BaseScenario
from .policy_bfs import ezpolicy, get_action_BFS  # noqa: F401

__all__ = ["make_env", "MultiAgentEnv", "ezpolicy", "get_action_BFS"]

_counter = [0]


def load_scenario(scenario_name):
    """Load `envs/<scenario_name>.py` (or an explicit path to a scenario file) and
    instantiate its `Scenario` - the reference's plugin mechanism (__init__.py:8-9)."""
    if osp.isfile(scenario_name):
        pathname = scenario_name
    else:
        pathname = osp.join(osp.dirname(__file__), 'envs', scenario_name + '.py')
    if not osp.isfile(pathname):
        raise FileNotFoundError(pathname)
    _counter[0] += 1
    spec = importlib.util.spec_from_file_location("formation_gym_scenario_%d" % _counter[0], pathname)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    cls = module.Scenario
    if getattr(cls, "step_batch", BaseScenario.step_batch) is BaseScenario.step_batch:
        # a file written for the reference (make_world / reset_world / observation / reward per agent, scenario.py:4-12):
        # physics on the GPU, its callbacks on the host - the slow path, see callback_scenario.py
        from .callback_scenario import CallbackScenario
        return CallbackScenario(cls)
    return cls()


def make_env(scenario_name='basic_formation_env', benchmark=False, num_agents=3,
             num_envs=1, device=None, **scenario_kwargs):
    """Reference signature `make_env(scenario_name, benchmark, num_agents)` plus the
    batch extension: `num_envs` independent environments on `device`."""
    scenario = load_scenario(scenario_name)
    world = scenario.make_world(num_agents, num_envs=num_envs, device=device, **scenario_kwargs)
    if benchmark:
        env = MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation,
                            scenario.benchmark_data, shared_viewer=True)
    else:
        env = MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation,
                            shared_viewer=True)
    env.info = {"path": getattr(scenario, "PATH", "fused HIP launch per step (batched Scenario protocol)")}
    return env
