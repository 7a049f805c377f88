// az_envs.h -- the three built-in environments as device functions.
//
// Native state: four floats + the step counter (CartPole: x, x_dot, theta,
// theta_dot; MountainCar: position, velocity, 0, 0; Acrobot: joint_angle1,
// joint_angle2, velocity_1, velocity_2).  Arithmetic is binary32 in the order the
// reference's step_env functions write it, on the elementary functions of az_math.h.
#ifndef AZ_ENVS_H_
#define AZ_ENVS_H_

#include "az_device.h"

struct EnvState {
  float s[4];
  int32_t time;
};

struct EnvOut {
  float obs[6];
  float reward, terminated, truncated;
};

__host__ __device__ inline int az_env_obs_dim(int env_id) {
  return env_id == AZ_ENV_CARTPOLE ? 4 : env_id == AZ_ENV_MOUNTAINCAR ? 2 : 6;
}
__host__ __device__ inline int az_env_num_actions(int env_id) {
  return env_id == AZ_ENV_CARTPOLE ? 2 : 3;
}

// gymnax/gymnax/environments/classic_control/cartpole.py:114-135
AZ_DEV void az_cartpole_is_terminal(const EnvState& st, float& terminated, float& truncated) {
  const float x_threshold = 2.4f, theta_threshold = 0.20943951023931953f;
  bool done1 = (st.s[0] < -x_threshold) || (st.s[0] > x_threshold);
  bool done2 = (st.s[2] < -theta_threshold) || (st.s[2] > theta_threshold);
  terminated = (done1 || done2) ? 1.0f : 0.0f;
  truncated = st.time >= 50000 ? 1.0f - terminated : 0.0f;
}

// cartpole.py:49-90
AZ_DEV void az_cartpole_step(EnvState& st, int action, EnvOut& o) {
  const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f,
              polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
  float prev_terminal, prev_truncated;
  az_cartpole_is_terminal(st, prev_terminal, prev_truncated);
  float x = st.s[0], x_dot = st.s[1], theta = st.s[2], theta_dot = st.s[3];
  float force = force_mag * (float)action - force_mag * (float)(1 - action);
  float costheta = az_cosf(theta), sintheta = az_sinf(theta);
  float temp = (force + (polemass_length * (theta_dot * theta_dot)) * sintheta) / total_mass;
  float thetaacc = (gravity * sintheta - costheta * temp) /
                   (length * (1.3333333333333333f - (masspole * (costheta * costheta)) / total_mass));
  float xacc = temp - ((polemass_length * thetaacc) * costheta) / total_mass;
  st.s[0] = x + tau * x_dot;
  st.s[1] = x_dot + tau * xacc;
  st.s[2] = theta + tau * theta_dot;
  st.s[3] = theta_dot + tau * thetaacc;
  st.time = st.time + 1;
  o.reward = 1.0f - az_maxf(prev_terminal, prev_truncated);
  az_cartpole_is_terminal(st, o.terminated, o.truncated);
#pragma unroll
  for (int i = 0; i < 4; ++i) o.obs[i] = st.s[i];
}

// mountain_car.py:43-72, 86-94
AZ_DEV void az_mountaincar_step(EnvState& st, int action, EnvOut& o) {
  const float min_position = -1.2f, max_position = 0.6f, max_speed = 0.07f, goal_position = 0.5f,
              goal_velocity = 0.0f, force = 0.001f, gravity = 0.0025f;
  float position = st.s[0], velocity = st.s[1];
  velocity = (velocity + (float)(action - 1) * force) - az_cosf(3.0f * position) * gravity;
  velocity = az_minf(az_maxf(velocity, -max_speed), max_speed);
  position = position + velocity;
  position = az_minf(az_maxf(position, min_position), max_position);
  int stop = (position == min_position) && (velocity < 0.0f);
  velocity = velocity * (float)(1 - stop);
  st.s[0] = position;
  st.s[1] = velocity;
  st.s[2] = 0.0f;
  st.s[3] = 0.0f;
  st.time = st.time + 1;
  o.reward = -1.0f;
  o.terminated = ((position >= goal_position) && (velocity >= goal_velocity)) ? 1.0f : 0.0f;
  o.truncated = st.time >= 5000 ? 1.0f : 0.0f;
  o.obs[0] = position;
  o.obs[1] = velocity;
}

// The value thread Q of the caller's group of four (threads 4k ... 4k + 3 of the wave) holds: one
// DPP move.  All four threads must be active.
template <int Q>
AZ_DEV float az_quad_value(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), Q * 0x55, 0xf, 0xf, true));
}

// Four sines / cosines at once.  QUAD: the caller is a group of four threads (4k ... 4k + 3 of the
// wave, all active) that hold the SAME arguments and all want all four results; thread q
// evaluates the q-th function (a sine and a cosine are one instruction stream, az_trig_) and the
// results are exchanged, so the group pays for one evaluation instead of four.  Each value is
// the same az_sinf / az_cosf of the same argument either way.
template <bool QUAD>
AZ_DEV void az_trig4(float x0, bool cos0, float x1, bool cos1, float x2, bool cos2, float x3, bool cos3,
                     float& y0, float& y1, float& y2, float& y3) {
  if (QUAD) {
    const int q = threadIdx.x & 3;
    const float x = q == 0 ? x0 : (q == 1 ? x1 : (q == 2 ? x2 : x3));
    const bool c = q == 0 ? cos0 : (q == 1 ? cos1 : (q == 2 ? cos2 : cos3));
    const float y = az_trig_(x, c);
    y0 = az_quad_value<0>(y);
    y1 = az_quad_value<1>(y);
    y2 = az_quad_value<2>(y);
    y3 = az_quad_value<3>(y);
  } else {
    y0 = az_trig_(x0, cos0);
    y1 = az_trig_(x1, cos1);
    y2 = az_trig_(x2, cos2);
    y3 = az_trig_(x3, cos3);
  }
}

// acrobot.py:202-233
template <bool QUAD = false>
AZ_DEV void az_acrobot_dsdt(const float (&s)[5], float (&out)[5]) {
  const float m1 = 1.0f, m2 = 1.0f, l1 = 1.0f, lc1 = 0.5f, lc2 = 0.5f, I1 = 1.0f, I2 = 1.0f,
              g = 9.8f, half_pi = 1.5707963267948966f;
  float a = s[4];
  float theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
  float c2, s2, c12, c1;  // cos(theta2), sin(theta2), cos(theta1 + theta2 - pi/2), cos(theta1 - pi/2)
  az_trig4<QUAD>(theta2, true, theta2, false, (theta1 + theta2) - half_pi, true, theta1 - half_pi, true,
                 c2, s2, c12, c1);
  float d1 = ((m1 * (lc1 * lc1) + m2 * (((l1 * l1) + (lc2 * lc2)) + (((2.0f * l1) * lc2) * c2))) + I1) + I2;
  float d2 = m2 * ((lc2 * lc2) + (l1 * lc2) * c2) + I2;
  float phi2 = ((m2 * lc2) * g) * c12;
  float phi1 = ((((((-m2) * l1) * lc2) * (dtheta2 * dtheta2)) * s2 -
                 (((((2.0f * m2) * l1) * lc2) * dtheta2) * dtheta1) * s2) +
                ((m1 * lc1 + m2 * l1) * g) * c1) +
               phi2;
  float ddtheta2 = (((a + (d2 / d1) * phi1) - (((m2 * l1) * lc2) * (dtheta1 * dtheta1)) * s2) - phi2) /
                   ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1);
  float ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
  out[0] = dtheta1;
  out[1] = dtheta2;
  out[2] = ddtheta1;
  out[3] = ddtheta2;
  out[4] = 0.0f;
}

// acrobot.py:236-247
AZ_DEV float az_acrobot_wrap(float x, float m, float M) {
  float diff = M - m;
  float go_up = x < m ? 1.0f : 0.0f;
  float go_down = x >= M ? 1.0f : 0.0f;
  float how_often = go_up * __builtin_ceilf((m - x) / diff) +
                    go_down * __builtin_floorf((x - M) / diff + 1.0f);
  return (x - (how_often * diff) * go_down) + (how_often * diff) * go_up;
}

// acrobot.py:52-106, 137-147, 250-258
// QUAD: called by a group of four threads with the same state and action (see az_trig4); all
// four end up with the same results.
template <bool QUAD = false>
AZ_DEV void az_acrobot_step(EnvState& st, int action, EnvOut& o) {
  const float dt = 0.2f, pi = 3.141592653589793f, max_vel_1 = 12.566370614359172f,
              max_vel_2 = 28.274333882308138f;
  float torque = (float)(action - 1);
  torque = torque + 0.0f;
  float y0[5] = {st.s[0], st.s[1], st.s[2], st.s[3], torque};
  float k1[5], k2[5], k3[5], k4[5], tmp[5];
  float dt2 = dt / 2.0f;
  az_acrobot_dsdt<QUAD>(y0, k1);
#pragma unroll
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k1[i];
  az_acrobot_dsdt<QUAD>(tmp, k2);
#pragma unroll
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt2 * k2[i];
  az_acrobot_dsdt<QUAD>(tmp, k3);
#pragma unroll
  for (int i = 0; i < 5; ++i) tmp[i] = y0[i] + dt * k3[i];
  az_acrobot_dsdt<QUAD>(tmp, k4);
  float ns[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    ns[i] = y0[i] + (dt / 6.0f) * (((k1[i] + 2.0f * k2[i]) + 2.0f * k3[i]) + k4[i]);
  float ja1 = az_acrobot_wrap(ns[0], -pi, pi);
  float ja2 = az_acrobot_wrap(ns[1], -pi, pi);
  float v1 = az_minf(az_maxf(ns[2], -max_vel_1), max_vel_1);
  float v2 = az_minf(az_maxf(ns[3], -max_vel_2), max_vel_2);
  float cos1, sin1, cos2, sin2;
  az_trig4<QUAD>(ja1, true, ja1, false, ja2, true, ja2, false, cos1, sin1, cos2, sin2);
  int done_angle = ((-cos1) - az_cosf(ja2 + ja1)) > 1.0f;
  o.reward = -1.0f * (float)(1 - done_angle);
  st.s[0] = ja1;
  st.s[1] = ja2;
  st.s[2] = v1;
  st.s[3] = v2;
  st.time = st.time + 1;
  o.terminated = done_angle ? 1.0f : 0.0f;
  o.truncated = st.time >= 5000 ? 1.0f : 0.0f;
  o.obs[0] = cos1;
  o.obs[1] = sin1;
  o.obs[2] = cos2;
  o.obs[3] = sin2;
  o.obs[4] = v1;
  o.obs[5] = v2;
}

template <int ENV>
AZ_DEV void az_env_step_t(EnvState& st, int action, EnvOut& o) {
  if (ENV == AZ_ENV_CARTPOLE)
    az_cartpole_step(st, action, o);
  else if (ENV == AZ_ENV_MOUNTAINCAR)
    az_mountaincar_step(st, action, o);
  else
    az_acrobot_step(st, action, o);
}

// Every branch steps into an EnvOut of its own and copies all of it out in one order: when the
// branches end in stores to different elements of one array, the optimiser merges them into a
// single store through a selected address, and the array then lives in scratch memory.
AZ_DEV void az_env_out_copy(EnvOut& o, const EnvOut& t) {
#pragma unroll
  for (int i = 0; i < 6; ++i) o.obs[i] = t.obs[i];
  o.reward = t.reward;
  o.terminated = t.terminated;
  o.truncated = t.truncated;
}

AZ_DEV void az_env_step_rt(int env_id, EnvState& st, int action, EnvOut& o) {
  EnvOut t{};
  if (env_id == AZ_ENV_CARTPOLE) {
    az_cartpole_step(st, action, t);
    az_env_out_copy(o, t);
  } else if (env_id == AZ_ENV_MOUNTAINCAR) {
    az_mountaincar_step(st, action, t);
    az_env_out_copy(o, t);
  } else {
    az_acrobot_step(st, action, t);
    if (env_id == AZ_ENV_ACROBOT_NOTERM) t.terminated = 0.0f;  // acrobot_no_term.py:151-157
    az_env_out_copy(o, t);
  }
}

// The environments an action count admits (az_env_num_actions): with A known at compile time
// the other environments' code is not generated.
// QUAD: the caller is a group of four threads with the same state and action (az_trig4).
template <int A_EXACT, bool QUAD = false>
AZ_DEV void az_env_step_for_actions(int env_id, EnvState& st, int action, EnvOut& o) {
  if (A_EXACT == 2) {
    az_cartpole_step(st, action, o);
  } else if (A_EXACT == 3) {
    EnvOut t{};
    if (env_id == AZ_ENV_MOUNTAINCAR) {
      az_mountaincar_step(st, action, t);
      az_env_out_copy(o, t);
    } else {
      az_acrobot_step<QUAD>(st, action, t);
      if (env_id == AZ_ENV_ACROBOT_NOTERM) t.terminated = 0.0f;  // acrobot_no_term.py:151-157
      az_env_out_copy(o, t);
    }
  } else {
    az_env_step_rt(env_id, st, action, o);
  }
}

// `obs` has room for six values whatever the environment; those past its observation size are
// scratch.  All six are written on one path: branches that end in stores to different elements
// get their last stores merged into one store through a selected address, which this build's
// compiler has been seen to leave undefined on one of the paths.
AZ_DEV void az_env_obs(int env_id, const EnvState& st, float* obs) {
  float o[6] = {st.s[0], st.s[1], st.s[2], st.s[3], 0.0f, 0.0f};  // CartPole: the state; MountainCar: its first two
  if (env_id >= AZ_ENV_ACROBOT) {  // (both Acrobot variants)
    const float c1 = az_cosf(st.s[0]), s1 = az_sinf(st.s[0]);
    const float c2 = az_cosf(st.s[1]), s2 = az_sinf(st.s[1]);
    o[0] = c1;
    o[1] = s1;
    o[2] = c2;
    o[3] = s2;
    o[4] = st.s[2];
    o[5] = st.s[3];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) obs[i] = o[i];
}

// reset_env of the three environments (cartpole.py:93-108, mountain_car.py:74-80,
// acrobot.py:108-122): uniform() bits drawn from the key in the partitionable layout.
AZ_DEV void az_env_reset_one(int env_id, uint32_t k0, uint32_t k1, EnvState& st) {
  if (env_id == AZ_ENV_MOUNTAINCAR) {
    st.s[0] = az_uniform_from_bits(az_random_bits32(k0, k1, 0u), -0.6f, -0.4f);
    st.s[1] = 0.0f;
    st.s[2] = 0.0f;
    st.s[3] = 0.0f;
  } else {
    float lim = env_id == AZ_ENV_CARTPOLE ? 0.05f : 0.1f;
    for (int i = 0; i < 4; ++i)
      st.s[i] = az_uniform_from_bits(az_random_bits32(k0, k1, (uint32_t)i), -lim, lim);
  }
  st.time = 0;
}

#endif  // AZ_ENVS_H_
