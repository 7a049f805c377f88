// az_replay.hip -- the replay queues of the learner side on the GPU (include/az_replay.h).
// Part of libaz_mcts.so (gfx950 only); built with -ffp-contract=off.
//
// HBM-bound byte work: rows are a ring [cap, D] (no roll on insert), the priority and
// weight columns live in their own contiguous arrays (a sampling pass streams 4 bytes
// per row, not a cache line per row), and the prefix sums run in the fixed association
// order of az_replay.h so that the CPU checker reproduces every bit.
#include <hip/hip_runtime.h>

#include <mutex>

#include "az_host.h"
#include "az_math.h"
#include "../../include/az_replay.h"

static inline int az_scan_chunks(int cap) { return (cap + AZ_SCAN_CHUNK - 1) / AZ_SCAN_CHUNK; }

extern "C" int64_t az_replay_partial_floats(int32_t cap) {
  if (cap < 1) return -1;
  return 2 * (int64_t)az_scan_chunks(cap) + 8;
}

static int az_check_replay(const az_replay* r, const az_replay_state* st) {
  if (!r || !st) return az_fail(AZ_ERR_INVALID, "null replay descriptor or state");
  if (r->cap < 1 || r->D < 3) return az_fail(AZ_ERR_INVALID, "replay needs cap >= 1 and D >= 3");
  if (!r->rows || !r->priority || !r->weight || !r->cum || !r->partial || !r->owner)
    return az_fail(AZ_ERR_INVALID, "replay descriptor has null buffers");
  if (st->head < 0 || st->head >= r->cap || st->insert_position < 0 ||
      st->insert_position > r->cap || st->sample_position < 0)
    return az_fail(AZ_ERR_INVALID, "replay state out of range");
  return AZ_OK;
}

AZ_DEV int az_phys(int logical, int head, int cap) {
  const int p = head + logical;
  return p >= cap ? p - cap : p;
}

// ---------------------------------------------------------------------------
__global__ void k_replay_init(az_replay r) {
  const int64_t total = (int64_t)r.cap * r.D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    r.rows[i] = 0.0f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r.cap; i += gridDim.x * blockDim.x) {
    r.priority[i] = 0.0f;
    r.weight[i] = 0.0f;
    r.owner[i] = -1;
  }
}

extern "C" int az_replay_init(const az_replay* r, az_replay_state* st, void* stream) {
  if (!r || !st) return az_fail(AZ_ERR_INVALID, "null replay descriptor or state");
  st->head = st->insert_position = st->sample_position = 0;
  if (int rc = az_check_replay(r, st)) return rc;
  k_replay_init<<<dim3(1024), dim3(256), 0, (hipStream_t)stream>>>(*r);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// one workgroup of 64 threads per inserted row
__global__ void k_replay_insert(az_replay r, const float* __restrict__ update, int n, int pos,
                                int head) {
  const int row = blockIdx.x;
  const int p = az_phys(pos + row, head, r.cap);
  const float* src = update + (size_t)row * r.D;
  float* dst = r.rows + (size_t)p * r.D;
  for (int c = threadIdx.x; c < r.D; c += blockDim.x) dst[c] = src[c];
  if (threadIdx.x == 0) {
    r.priority[p] = src[r.D - 2];
    r.weight[p] = src[r.D - 1];
  }
}

extern "C" int az_replay_insert(const az_replay* r, az_replay_state* st, const float* update,
                                int32_t n, void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!update || n < 1) return az_fail(AZ_ERR_INVALID, "bad insert arguments");
  if (n > r->cap)
    return az_fail(AZ_ERR_INVALID,
                   "Trying to insert a batch of samples larger than the maximum replay size. "
                   "num_samples: %d, max replay size %d", n, r->cap);
  // replay_buffers.py:143-153.  roll <= 0 drops the -roll oldest rows: the ring's head
  // moves forward instead of the data moving backward.
  int position = st->insert_position;
  const int roll = r->cap - position - n < 0 ? r->cap - position - n : 0;
  st->head = (st->head - roll) % r->cap;
  position += roll;
  k_replay_insert<<<dim3(n), dim3(64), 0, (hipStream_t)stream>>>(*r, update, n, position, st->head);
  AZ_LAUNCH_CHECK();
  st->insert_position = (position + n) % (r->cap + 1);
  st->sample_position = st->sample_position + roll > 0 ? st->sample_position + roll : 0;
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// prefix sums in the contract's order.  The scanned value of logical row i is the masked
// priority (i < n_valid ? priority : 0), divided by *divisor when divisor is non-null.
AZ_DEV float az_scan_value(const az_replay& r, int i, int n_valid, int head, const float* divisor) {
  float v = 0.0f;
  if (i < r.cap && i < n_valid) v = r.priority[az_phys(i, head, r.cap)];
  if (divisor) v = v / *divisor;
  return v;
}

// A chunk's values pass through LDS: a thread's run is AZ_SCAN_T CONSECUTIVE rows, which read
// straight from memory is 64 separate lines per wave access; staged, the workgroup fetches whole
// lines (thread t takes rows t, t + 256, ...) and every thread then picks its run out of LDS.
// Element x of the chunk sits at x + x / AZ_SCAN_T, so that the threads' runs start in
// different banks.
#define AZ_SCAN_STAGE (AZ_SCAN_CHUNK + AZ_SCAN_CHUNK / AZ_SCAN_T)
AZ_DEV int az_stage_index(int x) { return x + x / AZ_SCAN_T; }

// The run of AZ_SCAN_T values of thread t in chunk k: inclusive sums left to right.  `vals`
// (AZ_SCAN_STAGE floats of LDS) keeps the chunk's values for the caller.  Workgroup-wide call.
AZ_DEV void az_scan_run(const az_replay& r, int chunk, int t, int n_valid, int head,
                        const float* divisor, float (&incl)[AZ_SCAN_T], float* vals) {
  const int chunk_base = chunk * AZ_SCAN_CHUNK;
#pragma unroll
  for (int k = 0; k < AZ_SCAN_T; ++k) {
    const int x = k * AZ_SCAN_B + t;
    vals[az_stage_index(x)] = az_scan_value(r, chunk_base + x, n_valid, head, divisor);
  }
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < AZ_SCAN_T; ++j) {
    const float v = vals[az_stage_index(t * AZ_SCAN_T + j)];
    s = j == 0 ? v : s + v;
    incl[j] = s;
  }
}

// Left-to-right exclusive sums of the workgroup's run totals (in LDS), by one thread:
// the order is the contract, and 256 dependent adds per 4096 rows are cheap.
AZ_DEV float az_scan_runs(float* tot, float* off, int t) {
  __syncthreads();
  if (t == 0) {
    // (sixteen totals are fetched ahead of their sixteen dependent additions: one LDS round trip
    // per addition made this loop three quarters of the scan kernels' time)
    float s = 0.0f;
#pragma unroll 1
    for (int i = 0; i < AZ_SCAN_B; i += 16) {
      float v[16], o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tot[i + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        o[j] = s;
        s = s + v[j];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) off[i + j] = o[j];
    }
    // chunk total as the contract defines it: last run offset + last run total
    tot[AZ_SCAN_B] = off[AZ_SCAN_B - 1] + tot[AZ_SCAN_B - 1];
  }
  __syncthreads();
  return off[t];
}

// The offset of chunk `chunk`: the chunk totals before it added up left to right (the contract's
// order) by one thread of the workgroup, out of a copy in LDS that all threads fetch at once.  For
// up to AZ_CHUNKS_IN_KERNEL chunks this replaces a launch of its own (k_scan_chunks) in front of
// every consumer; `stage` holds AZ_CHUNKS_IN_KERNEL floats.
#define AZ_CHUNKS_IN_KERNEL 1024
AZ_DEV float az_chunk_offset(const float* __restrict__ chunk_tot, int chunk, float* stage, int t, int nthreads) {
  for (int i = t; i < chunk; i += nthreads) stage[i] = chunk_tot[i];
  __syncthreads();
  if (t == 0) {
    float s = 0.0f;
    int i = 0;
#pragma unroll 1
    for (; i + 16 <= chunk; i += 16) {  // (operands fetched sixteen at a time, as in az_scan_runs)
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = stage[i + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) s = s + v[j];
    }
    for (; i < chunk; ++i) s = s + stage[i];
    stage[AZ_CHUNKS_IN_KERNEL] = s;
  }
  __syncthreads();
  return stage[AZ_CHUNKS_IN_KERNEL];
}

// pass A: chunk totals
__global__ __launch_bounds__(AZ_SCAN_B) void k_scan_totals(az_replay r, int n_valid, int head,
                                                           const float* divisor,
                                                           float* __restrict__ chunk_tot) {
  __shared__ __attribute__((aligned(16))) float tot[AZ_SCAN_B + 4], off[AZ_SCAN_B];
  __shared__ float vals[AZ_SCAN_STAGE];
  float incl[AZ_SCAN_T];
  az_scan_run(r, blockIdx.x, threadIdx.x, n_valid, head, divisor, incl, vals);
  tot[threadIdx.x] = incl[AZ_SCAN_T - 1];
  az_scan_runs(tot, off, threadIdx.x);
  if (threadIdx.x == 0) chunk_tot[blockIdx.x] = tot[AZ_SCAN_B];
}

// level 2: chunk offsets left to right (one thread); also the grand total, which by the
// contract is cum[cap - 1]... computed by pass B's formula for the last row
__global__ void k_scan_chunks(const float* __restrict__ chunk_tot, int nchunks,
                              float* __restrict__ chunk_off) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.0f;
    for (int i = 0; i < nchunks; ++i) {
      chunk_off[i] = s;
      s = s + chunk_tot[i];
    }
  }
}

#ifdef AZ_STAMP
// Diagnostic build only: cycle stamps of the last workgroup of k_scan_final.
__device__ long long az_replay_stamps[16];
#define AZ_RSTAMP(k) \
  do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) az_replay_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int az_debug_read_replay_stamps(long long* host_out) {
  AZ_HIP(hipDeviceSynchronize());
  AZ_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(az_replay_stamps), sizeof(long long) * 16));
  return AZ_OK;
}
#else
#define AZ_RSTAMP(k) do { } while (0)
#endif

// pass B: cum[i] = (chunk_off + run_off) + incl; optionally the importance weights
// 1 / (n_valid * p + 1e-10)^beta of every row with their maximum (as uint bits: the
// weights are positive), replay_buffers.py:329-330.
__global__ __launch_bounds__(AZ_SCAN_B) void k_scan_final(az_replay r, int n_valid, int head,
                                                          const float* divisor,
                                                          const float* __restrict__ chunk_off,
                                                          const float* __restrict__ chunk_tot,
                                                          float* __restrict__ cum, int with_weights,
                                                          float beta, unsigned int* wmax_bits) {
  __shared__ __attribute__((aligned(16))) float tot[AZ_SCAN_B + 4], off[AZ_SCAN_B];
  __shared__ __attribute__((aligned(16))) float stage[AZ_CHUNKS_IN_KERNEL + 4];
  __shared__ float vals[AZ_SCAN_STAGE], cums[AZ_SCAN_STAGE];
  __shared__ unsigned int smax;
  float incl[AZ_SCAN_T];
  const int t = threadIdx.x;
  if (t == 0) smax = 0u;
  AZ_RSTAMP(0);
  az_scan_run(r, blockIdx.x, t, n_valid, head, divisor, incl, vals);
  AZ_RSTAMP(1);
  tot[t] = incl[AZ_SCAN_T - 1];
  const float run_off = az_scan_runs(tot, off, t);
  AZ_RSTAMP(2);
  // (chunk_tot != null: the chunk totals; the offsets are worked out here)
  const float c_off = chunk_tot ? az_chunk_offset(chunk_tot, blockIdx.x, stage, t, AZ_SCAN_B) : chunk_off[blockIdx.x];
  AZ_RSTAMP(3);
  const float base_off = c_off + run_off;
#pragma unroll
  for (int j = 0; j < AZ_SCAN_T; ++j) cums[az_stage_index(t * AZ_SCAN_T + j)] = base_off + incl[j];
  __syncthreads();
  AZ_RSTAMP(4);
  // out through LDS again: whole lines of cum and of the weights
  unsigned int mx = 0u;
#pragma unroll
  for (int k = 0; k < AZ_SCAN_T; ++k) {
    const int x = k * AZ_SCAN_B + t, i = blockIdx.x * AZ_SCAN_CHUNK + x;
    if (i < r.cap) {
      cum[i] = cums[az_stage_index(x)];
      if (with_weights) {
        const float p = vals[az_stage_index(x)];
        const float w = 1.0f / az_powf((float)n_valid * p + 1e-10f, beta);
        r.weight[az_phys(i, head, r.cap)] = w;
        const unsigned int b = az_f2u(w);
        mx = b > mx ? b : mx;
      }
    }
  }
  AZ_RSTAMP(5);
  if (with_weights) {
    // wave maximum first, then one LDS atomic per wave and one global atomic per workgroup, none
    // of which hands a value back (a returning global atomic is a 2 us round trip at the very end
    // of every workgroup)
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned int other = (unsigned int)__shfl_xor((int)mx, o);
      mx = other > mx ? other : mx;
    }
    if ((t & 63) == 0) __hip_atomic_fetch_max(&smax, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_max(wmax_bits, smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  AZ_RSTAMP(6);
}

__global__ void k_weights_normalise(az_replay r, const unsigned int* __restrict__ wmax_bits) {
  const float wmax = az_u2f(*wmax_bits);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r.cap; i += gridDim.x * blockDim.x)
    r.weight[i] = r.weight[i] / wmax;
}

// Runs the scan of (masked priority [/ divisor]) into r.cum; partial[0..nc) chunk totals,
// partial[nc..2nc) chunk offsets.
static int az_scan_launch(const az_replay* r, const az_replay_state* st, const float* divisor,
                          int with_weights, float beta, unsigned int* wmax_bits, bool need_cum,
                          hipStream_t s) {
  const int nc = az_scan_chunks(r->cap);
  float* chunk_tot = r->partial;
  float* chunk_off = r->partial + nc;
  k_scan_totals<<<dim3(nc), dim3(AZ_SCAN_B), 0, s>>>(*r, st->insert_position, st->head, divisor,
                                                      chunk_tot);
  AZ_LAUNCH_CHECK();
  const bool in_kernel = nc <= AZ_CHUNKS_IN_KERNEL;  // the consumers add the chunk totals up themselves
  if (!in_kernel) {
    k_scan_chunks<<<dim3(1), dim3(64), 0, s>>>(chunk_tot, nc, chunk_off);
    AZ_LAUNCH_CHECK();
  }
  if (need_cum) {
    k_scan_final<<<dim3(nc), dim3(AZ_SCAN_B), 0, s>>>(*r, st->insert_position, st->head, divisor,
                                                       chunk_off, in_kernel ? chunk_tot : nullptr, r->cum,
                                                       with_weights, beta, wmax_bits);
    AZ_LAUNCH_CHECK();
  }
  return AZ_OK;
}

// total[0] = cum[cap - 1] of the scan whose chunk totals / offsets are in `partial`:
// the last row's formula needs its run, so it is recomputed by one workgroup.
// chunk_tot != null: the chunk totals, from which the last chunk's offset is worked out here.
// zero_word (or null): a word set to zero on the way (the weights' running maximum).
__global__ __launch_bounds__(AZ_SCAN_B) void k_scan_last(az_replay r, int n_valid, int head,
                                                         const float* divisor,
                                                         const float* __restrict__ chunk_off,
                                                         const float* __restrict__ chunk_tot,
                                                         float* __restrict__ total, unsigned int* zero_word) {
  __shared__ __attribute__((aligned(16))) float tot[AZ_SCAN_B + 4], off[AZ_SCAN_B];
  __shared__ __attribute__((aligned(16))) float stage[AZ_CHUNKS_IN_KERNEL + 4];
  __shared__ float vals[AZ_SCAN_STAGE];
  float incl[AZ_SCAN_T];
  const int chunk = (r.cap - 1) / AZ_SCAN_CHUNK, t = threadIdx.x;
  az_scan_run(r, chunk, t, n_valid, head, divisor, incl, vals);
  tot[t] = incl[AZ_SCAN_T - 1];
  const float run_off = az_scan_runs(tot, off, t);
  const float c_off = chunk_tot ? az_chunk_offset(chunk_tot, chunk, stage, t, AZ_SCAN_B) : chunk_off[chunk];
  const int within = (r.cap - 1) - chunk * AZ_SCAN_CHUNK;
  if (t == within / AZ_SCAN_T) total[0] = (c_off + run_off) + incl[within % AZ_SCAN_T];
  if (t == 0 && zero_word) *zero_word = 0u;
}

// ---------------------------------------------------------------------------
// jax.random.choice(key, n, (bs,), replace=True, p): r = cum[-1] * (1 - uniform(key)),
// index = searchsorted(cum, r) (left).
__global__ void k_draw_prioritized(az_replay r, uint32_t k0, uint32_t k1, int bs,
                                   int32_t* __restrict__ out_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs) return;
  const float u = az_uniform_from_bits(az_random_bits32(k0, k1, (uint32_t)i), 0.0f, 1.0f);
  const float target = r.cum[r.cap - 1] * (1.0f - u);
  int lo = 0, hi = r.cap;  // first index with cum[index] >= target
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (r.cum[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  out_idx[i] = lo;
}

// jax.random.randint(key, (bs,), minval, maxval), int32
__global__ void k_draw_uniform(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, int bs,
                               int minval, int maxval, int32_t* __restrict__ out_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs) return;
  const uint32_t hi = az_random_bits32(a0, a1, (uint32_t)i), lo = az_random_bits32(b0, b1, (uint32_t)i);
  const uint32_t span = maxval > minval ? (uint32_t)(maxval - minval) : 1u;
  const uint32_t m = 65536u % span;
  const uint32_t mult = (m * m) % span;
  uint32_t off = (hi % span) * mult + (lo % span);
  off = off % span;
  out_idx[i] = minval + (int32_t)off;
}

// one workgroup of 64 threads per sampled row; indices wrap like jnp.take(mode='wrap')
__global__ void k_replay_gather(az_replay r, int head, const int32_t* __restrict__ idx,
                                float* __restrict__ out) {
  int logical = idx[blockIdx.x] % r.cap;
  if (logical < 0) logical += r.cap;
  const int p = az_phys(logical, head, r.cap);
  const float* src = r.rows + (size_t)p * r.D;
  float* dst = out + (size_t)blockIdx.x * r.D;
  for (int c = threadIdx.x; c < r.D - 2; c += blockDim.x) dst[c] = src[c];
  if (threadIdx.x == 0) {
    dst[r.D - 2] = r.priority[p];
    dst[r.D - 1] = r.weight[p];
  }
}

// k_draw_prioritized and k_weights_normalise in one launch: both only need the finished prefix
// sums / weight maximum, neither reads what the other writes.  The first `draw_blocks`
// workgroups draw, the others normalise.
__global__ __launch_bounds__(256) void k_draw_and_normalise(az_replay r, uint32_t k0, uint32_t k1, int bs,
                                                            int32_t* __restrict__ out_idx, int draw_blocks,
                                                            const unsigned int* __restrict__ wmax_bits) {
  if ((int)blockIdx.x < draw_blocks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs) return;
    const float u = az_uniform_from_bits(az_random_bits32(k0, k1, (uint32_t)i), 0.0f, 1.0f);
    const float target = r.cum[r.cap - 1] * (1.0f - u);
    int lo = 0, hi = r.cap;  // first index with cum[index] >= target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (r.cum[mid] < target)
        lo = mid + 1;
      else
        hi = mid;
    }
    out_idx[i] = lo;
    return;
  }
  const float wmax = az_u2f(*wmax_bits);
  const int nb = gridDim.x - draw_blocks;
  for (int i = (blockIdx.x - draw_blocks) * blockDim.x + threadIdx.x; i < r.cap; i += nb * blockDim.x)
    r.weight[i] = r.weight[i] / wmax;
}

// The sampling pass on a descriptor whose scratch pointers (cum, partial) are set.
static int az_sample_prioritized_impl(const az_replay* r, const az_replay_state* st,
                                      const uint32_t sample_key[2], float beta, int32_t bs,
                                      int32_t* out_idx, float* out_batch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nc = az_scan_chunks(r->cap);
  float* total = r->partial + 2 * nc;  // sum of the masked priorities
  unsigned int* wmax = reinterpret_cast<unsigned int*>(r->partial + 2 * nc + 1);
  // probs /= sum(probs): the sum is the last element of the (unnormalised) prefix sums
  const bool in_kernel = nc <= AZ_CHUNKS_IN_KERNEL;
  if (int rc = az_scan_launch(r, st, nullptr, 0, 0.0f, nullptr, false, s)) return rc;
  k_scan_last<<<dim3(1), dim3(AZ_SCAN_B), 0, s>>>(*r, st->insert_position, st->head, nullptr,
                                                  r->partial + nc, in_kernel ? r->partial : nullptr, total, wmax);
  AZ_LAUNCH_CHECK();
  if (int rc = az_scan_launch(r, st, total, 1, beta, wmax, true, s)) return rc;
  // (the draw stays a binary search per thread: the prefix sums are not monotone to the last bit
  // across run boundaries -- heavy-tailed priorities give a few inversions of one unit in the last
  // place per 10^4 rows -- so only the same probe sequence finds the same index as the checker's
  // searchsorted; a 64-way search was four round trips instead of twenty but not identical)
  const int draw_blocks = (bs + 255) / 256;
  k_draw_and_normalise<<<dim3(draw_blocks + 1024), dim3(256), 0, s>>>(*r, sample_key[0], sample_key[1], bs,
                                                                      out_idx, draw_blocks, wmax);
  AZ_LAUNCH_CHECK();
  k_replay_gather<<<dim3(bs), dim3(64), 0, s>>>(*r, st->head, out_idx, out_batch);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

extern "C" int az_replay_sample_prioritized(const az_replay* r, const az_replay_state* st,
                                            const uint32_t sample_key[2], float beta, int32_t bs,
                                            int32_t* out_idx, float* out_batch, void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!sample_key || bs < 1 || !out_idx || !out_batch)
    return az_fail(AZ_ERR_INVALID, "bad sample arguments");
  return az_sample_prioritized_impl(r, st, sample_key, beta, bs, out_idx, out_batch, stream);
}

// ---------------------------------------------------------------------------
// The same sampling pass for callers that cannot allocate: its scratch (the cumulative
// probabilities and the chunk partial sums) lives in a device array of the library.
#define AZ_REPLAY_SCRATCH_FLOATS \
  (AZ_REPLAY_NOALLOC_MAX_CAP + 2 * (AZ_REPLAY_NOALLOC_MAX_CAP / AZ_SCAN_CHUNK + 1) + 8)
static __device__ __align__(256) float scratch[AZ_REPLAY_SCRATCH_FLOATS];

// Address of `scratch` on the current device.  In host code the array's name is only a handle:
// the address each GPU holds it at comes from hipGetSymbolAddress, once per device, with that
// device current.
#define AZ_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static float* g_scratch_ptr[AZ_MAX_DEVICES];

static int az_replay_scratch(hipStream_t stream, float** out) {
  int dev = -1;
  AZ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= AZ_MAX_DEVICES)
    return az_fail(AZ_ERR_UNSUPPORTED, "device ordinal %d outside [0,%d)", dev, AZ_MAX_DEVICES);
  if (stream != nullptr) {
    // the kernels run on the stream's device and must be handed THAT device's copy of the array
    hipDevice_t sdev = -1;
    AZ_HIP(hipStreamGetDevice(stream, &sdev));
    if ((int)sdev != dev)
      return az_fail(AZ_ERR_INVALID, "the stream belongs to device %d, the current device is %d", (int)sdev, dev);
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    AZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return az_fail(AZ_ERR_HIP, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return AZ_OK;
}

extern "C" int az_replay_sample_prioritized_noalloc(const az_replay* r, const az_replay_state* st,
                                                    const uint32_t sample_key[2], float beta,
                                                    int32_t bs, int32_t* out_idx, float* out_batch,
                                                    void* stream) {
  if (!r || !st) return az_fail(AZ_ERR_INVALID, "null replay descriptor or state");
  if (r->cap > AZ_REPLAY_NOALLOC_MAX_CAP)
    return az_fail(AZ_ERR_UNSUPPORTED,
                   "az_replay_sample_prioritized_noalloc holds scratch for at most %d rows, the queue has %d: "
                   "use az_replay_sample_prioritized with caller-provided scratch",
                   AZ_REPLAY_NOALLOC_MAX_CAP, r->cap);
  float* base = nullptr;
  if (int rc = az_replay_scratch((hipStream_t)stream, &base)) return rc;
  az_replay rr = *r;  // the caller's cum / partial are not used (they may be null)
  rr.cum = base;
  rr.partial = base + AZ_REPLAY_NOALLOC_MAX_CAP;
  if (int rc = az_check_replay(&rr, st)) return rc;
  if (!sample_key || bs < 1 || !out_idx || !out_batch)
    return az_fail(AZ_ERR_INVALID, "bad sample arguments");
  // every word of the scratch that the pass reads is written by it first (chunk totals and
  // offsets, cum[0, cap), the total) or cleared by it (the running maximum of the weights)
  return az_sample_prioritized_impl(&rr, st, sample_key, beta, bs, out_idx, out_batch, stream);
}

extern "C" int az_replay_sample_uniform(const az_replay* r, const az_replay_state* st,
                                        const uint32_t sample_key[2], int32_t bs, int32_t* out_idx,
                                        float* out_batch, void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!sample_key || bs < 1 || !out_idx || !out_batch)
    return az_fail(AZ_ERR_INVALID, "bad sample arguments");
  hipStream_t s = (hipStream_t)stream;
  uint32_t a0, a1, b0, b1;  // randint splits its key once more
  az_key_split(sample_key[0], sample_key[1], 0u, &a0, &a1);
  az_key_split(sample_key[0], sample_key[1], 1u, &b0, &b1);
  k_draw_uniform<<<dim3((bs + 255) / 256), dim3(256), 0, s>>>(a0, a1, b0, b1, bs, st->sample_position,
                                                              st->insert_position, out_idx);
  AZ_LAUNCH_CHECK();
  k_replay_gather<<<dim3(bs), dim3(64), 0, s>>>(*r, st->head, out_idx, out_batch);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// ---------------------------------------------------------------------------
// set_priorities with a defined winner for repeated indices: the highest batch position.
__global__ void k_prio_claim(az_replay r, int head, const int32_t* __restrict__ idx, int bs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs) return;
  int logical = idx[i] % r.cap;
  if (logical < 0) logical += r.cap;
  atomicMax(&r.owner[az_phys(logical, head, r.cap)], i);
}
__global__ void k_prio_write(az_replay r, int head, const int32_t* __restrict__ idx,
                             const float* __restrict__ pri, int bs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs) return;
  int logical = idx[i] % r.cap;
  if (logical < 0) logical += r.cap;
  const int p = az_phys(logical, head, r.cap);
  if (r.owner[p] == i) r.priority[p] = pri[i];
}
__global__ void k_prio_release(az_replay r, int head, const int32_t* __restrict__ idx, int bs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs) return;
  int logical = idx[i] % r.cap;
  if (logical < 0) logical += r.cap;
  r.owner[az_phys(logical, head, r.cap)] = -1;
}

extern "C" int az_replay_set_priorities(const az_replay* r, const az_replay_state* st,
                                        const int32_t* idx, const float* priorities, int32_t bs,
                                        void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!idx || !priorities || bs < 1) return az_fail(AZ_ERR_INVALID, "bad set_priorities arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 g((bs + 255) / 256), b(256);
  k_prio_claim<<<g, b, 0, s>>>(*r, st->head, idx, bs);
  AZ_LAUNCH_CHECK();
  k_prio_write<<<g, b, 0, s>>>(*r, st->head, idx, priorities, bs);
  AZ_LAUNCH_CHECK();
  k_prio_release<<<g, b, 0, s>>>(*r, st->head, idx, bs);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

// ---------------------------------------------------------------------------
__global__ void k_replay_export(az_replay r, int head, float* __restrict__ out) {
  const int logical = blockIdx.x;
  const int p = az_phys(logical, head, r.cap);
  const float* src = r.rows + (size_t)p * r.D;
  float* dst = out + (size_t)logical * r.D;
  for (int c = threadIdx.x; c < r.D - 2; c += blockDim.x) dst[c] = src[c];
  if (threadIdx.x == 0) {
    dst[r.D - 2] = r.priority[p];
    dst[r.D - 1] = r.weight[p];
  }
}

extern "C" int az_replay_export(const az_replay* r, const az_replay_state* st, float* out,
                                void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!out) return az_fail(AZ_ERR_INVALID, "null export buffer");
  k_replay_export<<<dim3(r->cap), dim3(64), 0, (hipStream_t)stream>>>(*r, st->head, out);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}

__global__ void k_copy_f32(const float* __restrict__ src, float* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int az_replay_scan(const az_replay* r, const az_replay_state* st, float* out_cum,
                              float* out_total, void* stream) {
  if (int rc = az_check_replay(r, st)) return rc;
  if (!out_cum || !out_total) return az_fail(AZ_ERR_INVALID, "null scan outputs");
  hipStream_t s = (hipStream_t)stream;
  const int nc = az_scan_chunks(r->cap);
  if (int rc = az_scan_launch(r, st, nullptr, 0, 0.0f, nullptr, true, s)) return rc;
  k_scan_last<<<dim3(1), dim3(AZ_SCAN_B), 0, s>>>(*r, st->insert_position, st->head, nullptr,
                                                  r->partial + nc, nc <= AZ_CHUNKS_IN_KERNEL ? r->partial : nullptr,
                                                  out_total, nullptr);
  AZ_LAUNCH_CHECK();
  k_copy_f32<<<dim3(256), dim3(256), 0, s>>>(r->cum, out_cum, r->cap);
  AZ_LAUNCH_CHECK();
  return AZ_OK;
}
