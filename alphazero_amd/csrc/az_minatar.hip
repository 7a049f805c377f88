// az_minatar.hip -- MinAtar Breakout (include/az_minatar.h): integer game logic, one thread per
// lane; observations written by one thread per grid cell.
#include <hip/hip_runtime.h>

#include "../../include/az_minatar.h"
#include "az_host.h"

namespace {

struct Breakout {
  int ball_y, ball_x, ball_dir, pos, strike, last_y, last_x, time, terminal;
  uint32_t rows[10];
};

AZ_DEV void bk_load(const int32_t* w, Breakout& s) {
  s.ball_y = w[0]; s.ball_x = w[1]; s.ball_dir = w[2]; s.pos = w[3]; s.strike = w[4];
  s.last_y = w[5]; s.last_x = w[6]; s.time = w[7]; s.terminal = w[8];
  for (int r = 0; r < 10; ++r) s.rows[r] = (uint32_t)w[10 + r];
}
AZ_DEV void bk_store(int32_t* w, const Breakout& s) {
  w[0] = s.ball_y; w[1] = s.ball_x; w[2] = s.ball_dir; w[3] = s.pos; w[4] = s.strike;
  w[5] = s.last_y; w[6] = s.last_x; w[7] = s.time; w[8] = s.terminal; w[9] = 0;
  for (int r = 0; r < 10; ++r) w[10 + r] = (int32_t)s.rows[r];
}

// breakout.py:92-109.  jax.random.choice(key, [0, 1], shape=()) is randint(key, (), 0, 2):
// the key is split once more and, with a span of two, only the low word's parity survives.
AZ_DEV void bk_reset(uint32_t k0, uint32_t k1, Breakout& s) {
  uint32_t b0, b1;
  az_key_split(k0, k1, 1u, &b0, &b1);
  const int start = (int)(az_random_bits32(b0, b1, 0u) & 1u);
  s.ball_y = 3;
  s.ball_x = start ? 9 : 0;
  s.ball_dir = start ? 3 : 2;
  s.pos = 4;
  s.strike = 0;
  s.last_y = 3;
  s.last_x = s.ball_x;
  s.time = 0;
  s.terminal = 0;
  for (int r = 0; r < 10; ++r) s.rows[r] = (r >= 1 && r <= 3) ? 0x3ffu : 0u;
}

// brick_map[y, x] as a gather: indices past the map clamp to its edge
AZ_DEV int bk_brick(const Breakout& s, int y, int x) {
  y = y < 0 ? 0 : (y > 9 ? 9 : y);
  x = x < 0 ? 0 : (x > 9 ? 9 : x);
  return (int)((s.rows[y] >> x) & 1u);
}

// breakout.py:67-90, 168-285.  `action` indexes the minimal action set [0, 1, 3].
AZ_DEV void bk_step(Breakout& s, int action, int max_steps, float& reward, float& terminated,
                    float& truncated) {
  const int a = action == 1 ? 1 : (action == 2 ? 3 : 0);
  // step_agent (:168-214)
  const int old_pos = s.pos, old_x = s.ball_x, old_y = s.ball_y, old_dir = s.ball_dir;
  int pos = old_pos;
  if (a == 1) pos = old_pos - 1 > 0 ? old_pos - 1 : 0;
  if (a == 3) pos = old_pos + 1 < 9 ? old_pos + 1 : 9;
  int new_x = (old_dir == 1 || old_dir == 2) ? old_x + 1 : old_x - 1;
  int new_y = (old_dir == 2 || old_dir == 3) ? old_y + 1 : old_y - 1;
  int dir = old_dir;
  if (new_x < 0 || new_x > 9) {
    new_x = new_x < 0 ? 0 : 9;
    const int flip_x[4] = {1, 0, 3, 2};
    dir = flip_x[old_dir];
  }
  s.pos = pos;
  s.last_x = old_x;
  s.last_y = old_y;
  // step_ball_brick (:217-285); its `state` is the one step_agent returned: pos is the NEW
  // paddle position, ball_x / ball_y / strike still the old ones
  const int flip_y[4] = {3, 2, 1, 0}, flip_xy[4] = {2, 3, 0, 1};
  const bool border_y = new_y < 0;
  if (border_y) {
    new_y = 0;
    dir = flip_y[dir];
  }
  const bool strike_toggle = !border_y && bk_brick(s, new_y, new_x) == 1;
  const bool strike_bool = !s.strike && strike_toggle;
  reward = strike_bool ? 1.0f : 0.0f;
  if (strike_bool) {
    if (new_y >= 0 && new_y <= 9) s.rows[new_y] &= ~(1u << new_x);  // (a write past the map is dropped)
    new_y = s.last_y;
    dir = flip_y[dir];
  }
  const bool brick_cond = !strike_toggle && new_y == 9;
  bool empty = true;
  for (int r = 0; r < 10; ++r) empty = empty && s.rows[r] == 0u;
  if (brick_cond && empty)
    for (int r = 1; r <= 3; ++r) s.rows[r] = 0x3ffu;
  const bool redirect1 = brick_cond && old_x == s.pos;
  if (redirect1) {
    dir = flip_y[dir];
    new_y = s.last_y;
  }
  const bool redirect2 = brick_cond && !redirect1 && new_x == s.pos;
  if (redirect2) {
    dir = flip_xy[dir];
    new_y = s.last_y;
  }
  s.terminal = brick_cond && !redirect1 && !redirect2 ? 1 : 0;
  s.strike = strike_toggle ? 1 : 0;
  s.ball_dir = dir;
  s.ball_x = new_x;
  s.ball_y = new_y;
  s.time += 1;
  // is_terminal (:121-128)
  terminated = s.terminal ? 1.0f : 0.0f;
  truncated = s.time >= max_steps ? 1.0f - terminated : 0.0f;
}

// get_obs (:111-119): one thread per cell writes its four channels; positions outside the
// grid set nothing.
__global__ void k_breakout_obs(int B, const int32_t* __restrict__ state, float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 100) return;
  const int b = i / 100, cell = i % 100, y = cell / 10, x = cell % 10;
  const int32_t* w = state + (size_t)b * AZ_BREAKOUT_WORDS;
  float4 o;
  o.x = (y == 9 && x == w[3]) ? 1.0f : 0.0f;
  o.y = (y == w[0] && x == w[1]) ? 1.0f : 0.0f;
  o.z = (y == w[5] && x == w[6]) ? 1.0f : 0.0f;
  o.w = (float)(((uint32_t)w[10 + y] >> x) & 1u);
  reinterpret_cast<float4*>(obs)[i] = o;
}

__global__ void k_breakout_reset(int B, const uint32_t* __restrict__ keys, int32_t* __restrict__ state,
                                 uint32_t* __restrict__ rng_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Breakout s;
  bk_reset(keys[2 * b], keys[2 * b + 1], s);
  bk_store(state + (size_t)b * AZ_BREAKOUT_WORDS, s);
  if (rng_out) {  // brax.py:53
    uint32_t r0, r1;
    az_key_split(keys[2 * b], keys[2 * b + 1], 0u, &r0, &r1);
    rng_out[2 * b] = r0;
    rng_out[2 * b + 1] = r1;
  }
}

__global__ void k_breakout_step(int B, int max_steps, const int32_t* __restrict__ state_in,
                                const int32_t* __restrict__ action, int32_t* __restrict__ state_out,
                                float* __restrict__ reward, float* __restrict__ terminated,
                                float* __restrict__ truncated) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Breakout s;
  bk_load(state_in + (size_t)b * AZ_BREAKOUT_WORDS, s);
  float r, te, tr;
  bk_step(s, action[b], max_steps, r, te, tr);
  bk_store(state_out + (size_t)b * AZ_BREAKOUT_WORDS, s);
  reward[b] = r;
  terminated[b] = te;
  truncated[b] = tr;
}

// envs/brax_wrappers.py:81-97 (EpisodeWrapper), :109-134 (AutoResetWrapper) around brax.py:57-70;
// `pre` receives the state before the reset, whose observation is real_obs.
__global__ void k_breakout_real_step(int B, int episode_length, int max_steps, int32_t* __restrict__ state,
                                     int32_t* __restrict__ pre, float* __restrict__ reward,
                                     float* __restrict__ done, float* __restrict__ truncation,
                                     float* __restrict__ steps, uint32_t* __restrict__ rng,
                                     const int32_t* __restrict__ action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float st_steps = done[b] != 0.0f ? 0.0f : steps[b];
  uint32_t r0, r1;
  az_key_split(rng[2 * b], rng[2 * b + 1], 0u, &r0, &r1);
  rng[2 * b] = r0;
  rng[2 * b + 1] = r1;
  Breakout s;
  bk_load(state + (size_t)b * AZ_BREAKOUT_WORDS, s);
  float r, te, tr;
  bk_step(s, action[b], max_steps, r, te, tr);
  st_steps = st_steps + 1.0f;
  const float dn = st_steps >= (float)episode_length ? 1.0f : te;
  truncation[b] = st_steps >= (float)episode_length ? 1.0f - te : 0.0f;
  steps[b] = st_steps;
  reward[b] = r;
  done[b] = dn;
  bk_store(pre + (size_t)b * AZ_BREAKOUT_WORDS, s);
  if (dn != 0.0f) bk_reset(r0, r1, s);
  bk_store(state + (size_t)b * AZ_BREAKOUT_WORDS, s);
}

int launch_obs(int B, const int32_t* state, float* obs, hipStream_t s) {
  k_breakout_obs<<<dim3((B * 100 + 255) / 256), dim3(256), 0, s>>>(B, state, obs);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

}  // namespace

extern "C" int az_breakout_reset(int32_t B, const uint32_t* keys, int32_t* state, float* obs,
                                 uint32_t* rng_out, void* stream) {
  if (B < 1 || !keys || !state || !obs) return az_fail(AZ_ERR_INVALID, "bad breakout reset arguments");
  k_breakout_reset<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(B, keys, state, rng_out);
  AZ_LAUNCH_CHECK();
  return launch_obs(B, state, obs, (hipStream_t)stream);
}

extern "C" int az_breakout_step(int32_t B, int32_t max_steps_in_episode, const int32_t* state_in,
                                const int32_t* action, int32_t* state_out, float* obs, float* reward,
                                float* terminated, float* truncated, void* stream) {
  if (B < 1 || !state_in || !action || !state_out || !obs || !reward || !terminated || !truncated)
    return az_fail(AZ_ERR_INVALID, "bad breakout step arguments");
  k_breakout_step<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      B, max_steps_in_episode, state_in, action, state_out, reward, terminated, truncated);
  AZ_LAUNCH_CHECK();
  return launch_obs(B, state_out, obs, (hipStream_t)stream);
}

extern "C" int az_breakout_real_step(int32_t B, int32_t episode_length, int32_t max_steps_in_episode,
                                     int32_t* state, int32_t* pre_reset_state, float* obs, float* real_obs,
                                     float* reward, float* done, float* truncation, float* steps,
                                     uint32_t* rng, const int32_t* action, void* stream) {
  if (B < 1 || !state || !pre_reset_state || !obs || !real_obs || !reward || !done || !truncation ||
      !steps || !rng || !action)
    return az_fail(AZ_ERR_INVALID, "bad breakout real step arguments");
  k_breakout_real_step<<<dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      B, episode_length, max_steps_in_episode, state, pre_reset_state, reward, done, truncation, steps,
      rng, action);
  AZ_LAUNCH_CHECK();
  if (int rc = launch_obs(B, pre_reset_state, real_obs, (hipStream_t)stream)) return rc;
  return launch_obs(B, state, obs, (hipStream_t)stream);
}
