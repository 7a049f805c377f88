/* az_minatar.h -- C ABI of the MinAtar Breakout environment (libaz_mcts.so, gfx950).
 *
 * Replaces gymnax/gymnax/environments/minatar/breakout.py (MinBreakout: reset_env :92-109,
 * step_env :67-90 with step_agent :168-214 and step_ball_brick :217-285, get_obs :111-119,
 * is_terminal :121-128) under the wrappers the reference's driver puts around its
 * environments: GymnaxToBraxWrapper (gymnax/gymnax/wrappers/brax.py:41-70) for the model inside
 * the search, and EpisodeWrapper + AutoResetWrapper (envs/brax_wrappers.py:81-134) around it for
 * the real environment (alphazero.py:884-889).
 *
 * A lane's state is AZ_BREAKOUT_WORDS int32 words:
 *   0 ball_y  1 ball_x  2 ball_dir  3 pos  4 strike  5 last_y  6 last_x  7 time  8 terminal
 *   9 (unused)  10 .. 19 brick_map rows 0 .. 9, bit x of a word = brick_map[row, x]
 * Observations are float32 [B, 10, 10, 4] (channels paddle, ball, trail, brick).
 * All pointers are device pointers owned by the caller; calls enqueue on `stream` and do not
 * synchronise.  Return codes and az_last_error() as in az_mcts.h.
 */
#ifndef AZ_MINATAR_H_
#define AZ_MINATAR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_BREAKOUT_WORDS 20
#define AZ_BREAKOUT_OBS 400     /* 10 x 10 x 4 */
#define AZ_BREAKOUT_ACTIONS 3   /* the minimal action set ['n', 'l', 'r'] (breakout.py:53-58) */

/* GymnaxToBraxWrapper.reset (brax.py:41-55) on MinBreakout.reset_env (breakout.py:92-109):
 * keys [B,2] are the lanes' reset keys; rng_out [B,2] (may be NULL) receives info['_rng'] =
 * split(key)[0]. */
int az_breakout_reset(int32_t B, const uint32_t* keys, int32_t* state, float* obs, uint32_t* rng_out,
                      void* stream);

/* GymnaxToBraxWrapper.step (brax.py:57-70) on MinBreakout.step_env with max_steps_in_episode
 * (breakout.py:26: 1000): the model step of a search's recurrent_fn (alphazero.py:317-339).
 * state_in / state_out [B, AZ_BREAKOUT_WORDS] may be the same array; obs [B,400]; reward,
 * terminated, truncated [B] float32. */
int az_breakout_step(int32_t B, int32_t max_steps_in_episode, const int32_t* state_in,
                     const int32_t* action, int32_t* state_out, float* obs, float* reward,
                     float* terminated, float* truncated, void* stream);

/* One step of the real, auto-resetting environment (envs/brax_wrappers.py:81-97, 109-134 around
 * brax.py:57-70), in place: steps counts the episode's steps, done / truncation as the
 * EpisodeWrapper sets them, rng the lane's key; pre_reset_state [B, AZ_BREAKOUT_WORDS] receives
 * the stepped state before the reset and real_obs its observation. */
int az_breakout_real_step(int32_t B, int32_t episode_length, int32_t max_steps_in_episode,
                          int32_t* state, int32_t* pre_reset_state, float* obs, float* real_obs,
                          float* reward, float* done, float* truncation, float* steps, uint32_t* rng,
                          const int32_t* action, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AZ_MINATAR_H_ */
