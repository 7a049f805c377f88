/* az_replay.h -- C ABI of the GPU replay queues (libaz_mcts.so, gfx950).
 *
 * Replaces, for the learner side of the self-play loop, the reference's
 * replay_buffers.py: QueueBase.insert_internal (:120-159), UniformSamplingQueue
 * .sample_internal (:260-277), PrioritizedSamplingQueue.sample_internal (:304-335) and
 * .set_priorities (:337-339).  The reference keeps one flat f32 array data[cap, D] whose
 * last two columns are the priority and the importance weight (envs/transition.py:18-34,
 * alphazero.py:906-938), and physically rolls it when it is full; here the rows live in a
 * ring (logical row i is physical row (head + i) % cap), which gives the same logical
 * contents without moving cap*D floats per insert.  All indices exchanged through this
 * interface are LOGICAL indices, i.e. exactly the reference's row numbers.
 *
 * All pointers are device pointers owned by the caller; calls enqueue on `stream` and do
 * not synchronise.  Return codes and az_last_error() as in az_mcts.h.
 */
#ifndef AZ_REPLAY_H_
#define AZ_REPLAY_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Floating-point prefix sums are order dependent, so the order is part of the contract:
 * elements are taken in chunks of AZ_SCAN_CHUNK = AZ_SCAN_T * AZ_SCAN_B; inside a chunk
 * every run of AZ_SCAN_T consecutive elements is summed left to right, the AZ_SCAN_B run
 * totals of the chunk are summed left to right, the chunk totals are summed left to
 * right, and cum[i] = (chunk_offset + run_offset) + running_sum_in_run (binary32). */
#define AZ_SCAN_T 16
#define AZ_SCAN_B 256
#define AZ_SCAN_CHUNK (AZ_SCAN_T * AZ_SCAN_B)

typedef struct az_replay {
  float* rows;      /* [cap, D] ring of flat transitions (columns D-2, D-1 unused here)   */
  float* priority;  /* [cap] column D-2 of the reference layout, ring order               */
  float* weight;    /* [cap] column D-1 of the reference layout, ring order               */
  float* cum;       /* [cap] scratch: cumulative sampling probabilities, logical order    */
  float* partial;   /* [2 * ceil(cap / AZ_SCAN_CHUNK) + 8] scratch                        */
  int32_t* owner;   /* [cap] scratch of set_priorities, all -1 between calls              */
  int32_t cap, D;   /* D counts the two trailing columns                                  */
} az_replay;

/* ReplayBufferState (replay_buffers.py:58-66) minus the arrays; host memory. */
typedef struct az_replay_state {
  int32_t head;             /* physical row of logical row 0                              */
  int32_t insert_position;  /* replay_buffers.py:63                                       */
  int32_t sample_position;  /* replay_buffers.py:64                                       */
} az_replay_state;

/* Scratch sizes (elements) for a capacity. */
int64_t az_replay_partial_floats(int32_t cap);

/* Zeroes rows/priority/weight, sets owner to -1 and the state to empty
 * (QueueBase.init, replay_buffers.py:99-106). */
int az_replay_init(const az_replay* r, az_replay_state* st, void* stream);

/* QueueBase.insert_internal (replay_buffers.py:120-159): appends update[n, D]; when the
 * queue would overflow, the oldest rows are dropped first (the reference's roll).
 * AZ_ERR_INVALID when n > cap (check_can_insert, :108-118). */
int az_replay_insert(const az_replay* r, az_replay_state* st, const float* update, int32_t n,
                     void* stream);

/* PrioritizedSamplingQueue.sample_internal (replay_buffers.py:304-335):
 * probs = where(i < insert_position, priority, 0) / sum; draws `bs` logical indices with
 * replacement by inverse CDF (jax.random.choice: r = cum[-1] * (1 - uniform(sample_key)),
 * searchsorted left); rewrites EVERY row's weight as 1 / (insert_position * p + 1e-10)^beta
 * divided by the maximum over all cap rows; gathers the batch [bs, D] (with the fresh
 * weights).  sample_key = split(state.key)[1]; the caller keeps split(state.key)[0]. */
int az_replay_sample_prioritized(const az_replay* r, const az_replay_state* st,
                                 const uint32_t sample_key[2], float beta, int32_t bs,
                                 int32_t* out_idx, float* out_batch, void* stream);

/* No counterpart in the reference.  The same pass as az_replay_sample_prioritized for callers
 * that cannot allocate or own scratch memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): r->cum and r->partial are ignored (they may be
 * NULL) and the pass works in a device array inside the library instead, which holds scratch
 * for queues of at most AZ_REPLAY_NOALLOC_MAX_CAP rows (about 4 MiB per GPU the library runs
 * on, kept until the process ends); larger queues are refused with AZ_ERR_UNSUPPORTED.  The
 * call allocates nothing.  All calls on one GPU share that array: calls that overlap on one GPU
 * (different streams or threads) race.  `stream` must belong to the current device
 * (AZ_ERR_INVALID otherwise).  The first call on a device looks the array's address up
 * (hipGetSymbolAddress, which may load the code object): make that one call outside graph
 * capture; later calls on the device only launch kernels. */
#define AZ_REPLAY_NOALLOC_MAX_CAP (1 << 20)
int az_replay_sample_prioritized_noalloc(const az_replay* r, const az_replay_state* st,
                                         const uint32_t sample_key[2], float beta, int32_t bs,
                                         int32_t* out_idx, float* out_batch, void* stream);

/* UniformSamplingQueue.sample_internal (replay_buffers.py:260-277): indices =
 * jax.random.randint(sample_key, (bs,), sample_position, insert_position), rows gathered
 * with wrap-around (jnp.take mode='wrap'). */
int az_replay_sample_uniform(const az_replay* r, const az_replay_state* st,
                             const uint32_t sample_key[2], int32_t bs, int32_t* out_idx,
                             float* out_batch, void* stream);

/* PrioritizedSamplingQueue.set_priorities (replay_buffers.py:337-339):
 * priority[idx[i]] = priorities[i].  Where an index repeats, the last occurrence wins
 * (the reference leaves that order undefined). */
int az_replay_set_priorities(const az_replay* r, const az_replay_state* st, const int32_t* idx,
                             const float* priorities, int32_t bs, void* stream);

/* The reference's data[cap, D] (logical order, with the priority / weight columns). */
int az_replay_export(const az_replay* r, const az_replay_state* st, float* out, void* stream);

/* cum[] and the total of the masked priorities as the sampler computes them
 * (diagnostic / parity): out_cum[cap], out_total[1]. */
int az_replay_scan(const az_replay* r, const az_replay_state* st, float* out_cum,
                   float* out_total, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AZ_REPLAY_H_ */
