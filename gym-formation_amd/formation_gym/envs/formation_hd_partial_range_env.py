"""formation_hd_partial_range_env (reference envs/formation_hd_partial_range_env.py):
relative positions of the other agents are clipped to +-obs_range (:46).  MI355X-native plugin."""
from formation_gym import _native
from formation_gym.landmark_scenario import LandmarkScenario


class Scenario(LandmarkScenario):
    KIND = _native.FG_SCN_RANGE
    AGENT_SIZE = 0.04
    LANDMARK_SIZE = 0.02

    def make_world(self, num_agents=4, num_landmarks=4, obs_range=0.7, world_length=25, num_envs=1, device=None):
        self.obs_range = obs_range
        return self._build_world(num_agents, num_landmarks, 0, world_length, num_envs, device)
