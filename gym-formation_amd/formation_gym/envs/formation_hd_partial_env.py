"""formation_hd_partial_env (reference envs/formation_hd_partial_env.py): each agent
observes only its `num_obs` ring neighbours (:44-46).  MI355X-native plugin."""
from formation_gym import _native
from formation_gym.landmark_scenario import LandmarkScenario


class Scenario(LandmarkScenario):
    KIND = _native.FG_SCN_PARTIAL
    AGENT_SIZE = 0.04
    LANDMARK_SIZE = 0.02

    def make_world(self, num_agents=5, num_landmarks=5, num_obs=3, world_length=25, num_envs=1, device=None):
        self.num_obs = num_obs
        return self._build_world(num_agents, num_landmarks, 0, world_length, num_envs, device)
