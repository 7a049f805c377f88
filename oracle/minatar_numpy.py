"""CPU restatement of MinAtar Breakout (TEST INFRASTRUCTURE ONLY; the product never imports it).

Follows gymnax/gymnax/environments/minatar/breakout.py statement by statement on Python ints and
a NumPy brick map -- reset_env (:92-109), step_env (:67-90), step_agent (:168-214),
step_ball_brick (:217-285), get_obs (:111-119), is_terminal (:121-128) -- and the wrappers around
it: GymnaxToBraxWrapper (gymnax/gymnax/wrappers/brax.py:41-70), EpisodeWrapper and
AutoResetWrapper (envs/brax_wrappers.py:81-134).  jax.random.choice / randint / split are restated
from their public algorithm on the shared threefry (oracle/replay_numpy.py).

PARITY UNPINNED: the reference holds no recorded Breakout trajectories and its JAX runtime is
not installable here; the known-answer cases in tests/test_oracle_kats.py are derived by hand
from the source.
"""
import numpy as np

from . import c_oracle
from . import replay_numpy as rn

ACTION_SET = (0, 1, 3)   # minimal action set ['n', 'l', 'r'] (:53-58)


class EnvState:
    def __init__(self):
        self.ball_y = self.ball_x = self.ball_dir = self.pos = 0
        self.brick_map = np.zeros((10, 10), np.int32)
        self.strike = False
        self.last_y = self.last_x = 0
        self.time = 0
        self.terminal = False

    def copy(self):
        s = EnvState()
        s.__dict__.update(self.__dict__)
        s.brick_map = self.brick_map.copy()
        return s

    def words(self):
        """The packed form of include/az_minatar.h."""
        w = np.zeros(20, np.int32)
        w[:9] = [self.ball_y, self.ball_x, self.ball_dir, self.pos, int(self.strike), self.last_y,
                 self.last_x, self.time, int(self.terminal)]
        for r in range(10):
            w[10 + r] = sum(int(self.brick_map[r, x]) << x for x in range(10))
        return w

    @staticmethod
    def from_words(w):
        s = EnvState()
        (s.ball_y, s.ball_x, s.ball_dir, s.pos, strike, s.last_y, s.last_x, s.time, term) = (int(v) for v in w[:9])
        s.strike, s.terminal = bool(strike), bool(term)
        for r in range(10):
            for x in range(10):
                s.brick_map[r, x] = (int(w[10 + r]) >> x) & 1
        return s


def reset_env(key):
    """:92-109"""
    ball_start = int(rn.randint(key, 1, 0, 2)[0])   # jax.random.choice(key, [0, 1], shape=())
    s = EnvState()
    s.ball_y = 3
    s.ball_x = (0, 9)[ball_start]
    s.ball_dir = (2, 3)[ball_start]
    s.pos = 4
    s.brick_map[1:4, :] = 1
    s.strike = False
    s.last_y = 3
    s.last_x = (0, 9)[ball_start]
    s.time = 0
    s.terminal = False
    return s


def get_obs(s):
    """:111-119 (a scatter outside the grid is dropped)."""
    obs = np.zeros((10, 10, 4), np.float32)
    obs[9, s.pos, 0] = 1
    if 0 <= s.ball_y <= 9 and 0 <= s.ball_x <= 9:
        obs[s.ball_y, s.ball_x, 1] = 1
    if 0 <= s.last_y <= 9 and 0 <= s.last_x <= 9:
        obs[s.last_y, s.last_x, 2] = 1
    obs[:, :, 3] = s.brick_map
    return obs


def _gather(brick_map, y, x):
    return int(brick_map[min(max(y, 0), 9), min(max(x, 0), 9)])   # out-of-range gathers clamp


def step_agent(state, action):
    """:168-214"""
    s = state.copy()
    pos = (max(0, state.pos - 1) * (action == 1) + min(9, state.pos + 1) * (action == 3)
           + state.pos * (action != 1 and action != 3))
    last_x, last_y = state.ball_x, state.ball_y
    d = state.ball_dir
    new_x = ((state.ball_x - 1) * (d == 0) + (state.ball_x + 1) * (d == 1) + (state.ball_x + 1) * (d == 2)
             + (state.ball_x - 1) * (d == 3))
    new_y = ((state.ball_y - 1) * (d == 0) + (state.ball_y - 1) * (d == 1) + (state.ball_y + 1) * (d == 2)
             + (state.ball_y + 1) * (d == 3))
    border_cond_x = new_x < 0 or new_x > 9
    if border_cond_x:
        new_x = 0 * (new_x < 0) + 9 * (new_x > 9)
    ball_dir = (1, 0, 3, 2)[d] if border_cond_x else d
    s.pos, s.last_x, s.last_y, s.ball_dir = pos, last_x, last_y, ball_dir
    return s, new_x, new_y


def step_ball_brick(state, new_x, new_y):
    """:217-285"""
    reward = 0.0
    border_cond1_y = new_y < 0
    new_y = 0 if border_cond1_y else new_y
    ball_dir = (3, 2, 1, 0)[state.ball_dir] if border_cond1_y else state.ball_dir
    strike_toggle = (not border_cond1_y) and _gather(state.brick_map, new_y, new_x) == 1
    strike_bool = (not state.strike) and strike_toggle
    reward += strike_bool * 1.0
    brick_map = state.brick_map.copy()
    if strike_bool and 0 <= new_y <= 9 and 0 <= new_x <= 9:
        brick_map[new_y, new_x] = 0
    new_y = state.last_y if strike_bool else new_y
    ball_dir = (3, 2, 1, 0)[ball_dir] if strike_bool else ball_dir
    brick_cond = (not strike_toggle) and new_y == 9
    spawn_bricks = brick_cond and np.count_nonzero(brick_map) == 0
    if spawn_bricks:
        brick_map[1:4, :] = 1
    redirect_ball1 = brick_cond and state.ball_x == state.pos
    ball_dir = (3, 2, 1, 0)[ball_dir] if redirect_ball1 else ball_dir
    new_y = state.last_y if redirect_ball1 else new_y
    redirect_ball2 = brick_cond and (not redirect_ball1) and new_x == state.pos
    ball_dir = (2, 3, 0, 1)[ball_dir] if redirect_ball2 else ball_dir
    new_y = state.last_y if redirect_ball2 else new_y
    terminal = brick_cond and (not redirect_ball1) and (not redirect_ball2)
    s = state.copy()
    s.ball_dir, s.brick_map, s.strike = ball_dir, brick_map, bool(strike_toggle)
    s.ball_x, s.ball_y, s.terminal = new_x, new_y, bool(terminal)
    return s, np.float32(reward)


def step_env(state, action, max_steps_in_episode=1000):
    """:67-90; returns (obs, state, reward, terminated, truncated)."""
    a = ACTION_SET[int(action)]
    state, new_x, new_y = step_agent(state, a)
    state, reward = step_ball_brick(state, new_x, new_y)
    state.time = state.time + 1
    done = np.float32(state.terminal)
    done_steps = state.time >= max_steps_in_episode
    truncated = np.float32(1 - done if done_steps else 0.0)
    return get_obs(state), state, reward, done, truncated


class Wrapped:
    """One lane of AutoResetWrapper(EpisodeWrapper(GymnaxToBraxWrapper(env)))."""

    def __init__(self, key, episode_length=1000, max_steps_in_episode=1000):
        self.episode_length, self.max_steps = episode_length, max_steps_in_episode
        key = np.asarray(key, np.uint32)
        self.state = reset_env(key)
        self.rng = rn.split(key)[0]                      # brax.py:53
        self.obs = self.real_obs = get_obs(self.state)
        self.reward = self.done = self.truncation = self.steps = np.float32(0)

    def step(self, action):
        steps = np.float32(0) if self.done else self.steps        # brax_wrappers.py:113-116
        rng, _ = rn.split(self.rng)                               # brax.py:63 (the step key is unused)
        self.rng = rng
        obs, state, reward, terminated, _ = step_env(self.state, action, self.max_steps)
        steps = np.float32(steps + 1)
        done = np.float32(1) if steps >= self.episode_length else terminated     # :88-92
        self.truncation = np.float32(1 - terminated if steps >= self.episode_length else 0)
        self.steps, self.reward, self.done, self.real_obs = steps, reward, done, obs
        self.pre_reset_state = state
        if done:                                                  # :118-134
            state = reset_env(rng)
            obs = get_obs(state)
        self.state, self.obs = state, obs
        return self
