"""formation_hd_obs_env (reference envs/formation_hd_obs_env.py): falling obstacles that
are movable colliders of World.step (:36-42), -2 per collision (:92-98), obstacle velocity
re-armed by the reward callback (:84-89).  MI355X-native plugin."""
from formation_gym import _native
from formation_gym.landmark_scenario import LandmarkScenario


class Scenario(LandmarkScenario):
    KIND = _native.FG_SCN_OBSTACLE
    AGENT_SIZE = 0.1
    LANDMARK_SIZE = 0.02
    OBSTACLE_SIZE = 0.15
    PENALTY = 2.0

    def make_world(self, num_agents=4, num_landmarks=4, num_obstacles=3, world_length=50, num_envs=1, device=None):
        return self._build_world(num_agents, num_landmarks, num_obstacles, world_length, num_envs, device)
