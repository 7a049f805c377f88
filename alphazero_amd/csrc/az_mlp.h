// az_mlp.h -- policy / value MLP forward for a tile of 16 batch lanes on the
// f32-input matrix cores (v_mfma_f32_16x16x4_f32).
//
// Why f32 MFMA: its result is bit for bit the k-ordered fused-multiply-add chain
//   acc = fma(x[k], W[k][n], acc),  k = 0, 1, 2, ...
// which is exactly how the CPU checker evaluates a dense layer, so the fused
// search stays bit-identical to the host evaluation while the contraction runs
// on the matrix pipe instead of the vector ALU.
//
// Data layout.  A dense layer in -> out is padded to Kp = ceil64(in) and Np = ceil64(out)
// (Np = ceil16(out) for a network's last layer), so that the k-loop always runs whole
// chunks of four 16-row k-blocks; the padding is zeros, which leave an accumulator
// that started at +0 unchanged.
//  * Packed weights: for output tile t (16 columns) and k-block kb (16 rows) one
//    float4 per lane,  Wp[t][kb][lane][q] = W[16*kb + 4*q + (lane>>4)][16*t + (lane&15)],
//    i.e. the B operand of four consecutive MFMA k-steps; a wave reads 1 KiB
//    contiguous per load.  Zero padded.  Packed bias: Np floats, zero padded.
//  * Activations in LDS: row m (lane of the tile) holds its Kp inputs permuted as
//    [k & 3][k >> 2] so that the A operands of four consecutive k-steps are one
//    aligned float4:  act[m*stride + (k&3)*(Kp/4) + (k>>2)].
//
// Schedule.  Layer i of the policy network and layer i of the value network form
// phase i (one workgroup barrier per phase).  A phase's work items are pairs of
// output tiles; wave w takes items w, w + NWAVES, ...  With M = 16 every weight is
// used for only 16 multiply-adds, so the kernel is bound by streaming weights from
// L2, not by the matrix pipe alone: a wave holds the 16 k-blocks of its current unit
// in 128 VGPRs and refills each register slot, right after the MFMAs that consumed
// it, with the same k-block of its NEXT unit -- which may belong to the next phase,
// since weights do not depend on activations -- so up to 32 KiB per wave stay in
// flight behind the MFMAs, the epilogue and the barrier.
#ifndef AZ_MLP_H_
#define AZ_MLP_H_

#include "az_device.h"

typedef __attribute__((ext_vector_type(4))) float az_f32x4;


struct MlpLayout {
  int n_layers;
  int kin[AZ_MAX_LAYERS], nout[AZ_MAX_LAYERS];
  int kp[AZ_MAX_LAYERS], np[AZ_MAX_LAYERS];
  int64_t w_off[AZ_MAX_LAYERS], b_off[AZ_MAX_LAYERS];  // float offsets into the packed buffer
};

struct ModelDev {
  int env_id;
  float gamma;
  int obs_dim, A, K;
  int pol_kp, val_kp;  // widest padded layer of each network
  MlpLayout pol, val;
  const float* packed;
  const float* mean;
  const float* std;
};

__host__ __device__ inline int az_ceil16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline int az_ceil64(int x) { return (x + 63) & ~63; }
#define AZ_IN0_STRIDE 68  // row stride of the first layer's input buffer (Kp = 64, + 4)

// Fills `L` from the layer sizes; returns the offset just past this network.
inline int64_t az_mlp_layout(int n_layers, const int32_t* sizes, int64_t base, MlpLayout* L) {
  L->n_layers = n_layers;
  int64_t off = base;
  for (int i = 0; i < n_layers; ++i) {
    L->kin[i] = sizes[i];
    L->nout[i] = sizes[i + 1];
    L->kp[i] = az_ceil64(sizes[i]);
    L->np[i] = i == n_layers - 1 ? az_ceil16(sizes[i + 1]) : az_ceil64(sizes[i + 1]);
    L->w_off[i] = off;
    off += (int64_t)L->kp[i] * L->np[i];
    L->b_off[i] = off;
    off += L->np[i];
  }
  return off;
}

// Activation row stride (floats) for a network whose widest padded layer is kp:
// +4 floats shifts consecutive rows by one 16-byte slot, which keeps the float4
// A-operand reads of the 16 rows on distinct LDS banks.
__host__ __device__ inline int az_act_stride(int kp) { return kp + 4; }

AZ_DEV int az_act_index(int k, int kp) { return (k & 3) * (kp >> 2) + (k >> 2); }

// LDS floats of the network scratch: in0 [16][AZ_IN0_STRIDE], two ping-pong activation buffers
// per network, logits [16][AZ_MAX_ACTIONS], values [16][AZ_MAX_ATOMS].
__host__ __device__ inline int az_mlp_lds_floats(int pol_kp, int val_kp) {
  return 16 * AZ_IN0_STRIDE + 2 * 16 * az_act_stride(pol_kp) + 2 * 16 * az_act_stride(val_kp) +
         16 * AZ_MAX_ACTIONS + 16 * AZ_MAX_ATOMS;
}

struct MlpLds {
  float *in0, *pa, *pb, *va, *vb, *logits, *value;
  int pstride, vstride;
  AZ_DEV MlpLds(float* base, int pol_kp, int val_kp) {
    pstride = az_act_stride(pol_kp);
    vstride = az_act_stride(val_kp);
    in0 = base;
    pa = in0 + 16 * AZ_IN0_STRIDE;
    pb = pa + 16 * pstride;
    va = pb + 16 * pstride;
    vb = va + 16 * vstride;
    logits = vb + 16 * vstride;
    value = logits + 16 * AZ_MAX_ACTIONS;
  }
};

// One unit of work: a pair of output tiles of one layer of one network, restricted to
// one round of AZ_SLOTS k-blocks (layers wider than 16*AZ_SLOTS inputs take several
// rounds; the accumulators carry over).
#define AZ_SLOTS 16

struct MlpUnit {
  const az_f32x4* w0;  // packed weights of tile t0 for this round (+ lane), kb stride 64
  const az_f32x4* w1;  // tile t0 + 1, or w0 again when the pair is incomplete
  const float* bias;
  const float* ap;     // this lane's A operands for the round (LDS)
  float* out_act;      // next layer's activations (LDS), or null for the last layer
  float* out_plain;    // [16][out_stride] for the last layer
  int out_stride, out_np, nout, nkb, t0;
  int phase, idx, round;
  bool two, first, last, valid;
};

AZ_DEV int az_phase_items(const ModelDev& md, int phase, int& n_val) {
  n_val = phase < md.val.n_layers ? ((md.val.np[phase] >> 4) + 1) >> 1 : 0;
  const int n_pol = phase < md.pol.n_layers ? ((md.pol.np[phase] >> 4) + 1) >> 1 : 0;
  return n_val + n_pol;
}

AZ_DEV MlpUnit az_mlp_unit(const ModelDev& md, const MlpLds& s, int phase, int idx, int round,
                           int lane) {
  MlpUnit u;
  u.phase = phase;
  u.idx = idx;
  u.round = round;
  u.valid = true;
  int n_val;
  az_phase_items(md, phase, n_val);
  const bool is_val = idx < n_val;
  const MlpLayout& L = is_val ? md.val : md.pol;
  const int pair = is_val ? idx : idx - n_val;
  const int kp = L.kp[phase], np = L.np[phase];
  const int nkb_total = kp >> 4;
  const int kb_base = round * AZ_SLOTS;
  u.nkb = nkb_total - kb_base < AZ_SLOTS ? nkb_total - kb_base : AZ_SLOTS;
  u.first = round == 0;
  u.last = kb_base + AZ_SLOTS >= nkb_total;
  u.t0 = pair * 2;
  u.two = u.t0 + 1 < (np >> 4);
  const az_f32x4* wbase = reinterpret_cast<const az_f32x4*>(md.packed + L.w_off[phase]);
  u.w0 = wbase + ((size_t)u.t0 * nkb_total + kb_base) * 64 + lane;
  u.w1 = u.two ? u.w0 + (size_t)nkb_total * 64 : u.w0;
  u.bias = md.packed + L.b_off[phase];
  const int stride = is_val ? s.vstride : s.pstride;
  float* a = is_val ? s.va : s.pa;
  float* b = is_val ? s.vb : s.pb;
  // phase 0 reads in0; phase i > 0 reads the buffer phase i-1 wrote
  const float* in = phase == 0 ? s.in0 : ((phase & 1) ? a : b);
  const int in_stride = phase == 0 ? AZ_IN0_STRIDE : stride;
  u.ap = in + (lane & 15) * in_stride + (lane >> 4) * (kp >> 2) + kb_base * 4;
  const bool last_layer = phase == L.n_layers - 1;
  u.out_act = last_layer ? nullptr : ((phase & 1) ? b : a);
  u.out_plain = is_val ? s.value : s.logits;
  u.out_stride = last_layer ? (is_val ? AZ_MAX_ATOMS : AZ_MAX_ACTIONS) : stride;
  u.out_np = np;
  u.nout = L.nout[phase];
  return u;
}

// The unit wave `wave` runs after (phase, idx, round); `start` asks for its first unit.
template <int NWAVES>
AZ_DEV MlpUnit az_mlp_next_unit(const ModelDev& md, const MlpLds& s, int P, int phase, int idx,
                                int round, bool start, int wave, int lane) {
  if (!start) {
    const MlpLayout& L0 = md.val;  // rounds are a property of the item's layer width
    int n_val;
    const int n = az_phase_items(md, phase, n_val);
    const int kp = (idx < n_val ? L0.kp[phase] : md.pol.kp[phase]);
    const int nrounds = ((kp >> 4) + AZ_SLOTS - 1) / AZ_SLOTS;
    if (round + 1 < nrounds) return az_mlp_unit(md, s, phase, idx, round + 1, lane);
    if (idx + NWAVES < n) return az_mlp_unit(md, s, phase, idx + NWAVES, 0, lane);
    phase += 1;
  }
  for (; phase < P; ++phase) {
    int n_val;
    if (wave < az_phase_items(md, phase, n_val)) return az_mlp_unit(md, s, phase, wave, 0, lane);
  }
  MlpUnit none;
  none.valid = false;
  none.phase = P;
  none.nkb = 0;
  return none;
}

// Workgroup barrier that publishes LDS writes but leaves the weight prefetch (global
// loads into registers) in flight: a plain __syncthreads() would drain it.
AZ_DEV void az_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef AZ_STAMP
// Diagnostic build only (-DAZ_STAMP): cycle stamps of workgroup 0, one row per wave.
__device__ long long az_stamps[8][64];
#define AZ_STAMP_AT()                                                                 \
  do {                                                                                \
    if (blockIdx.x == 0 && lane == 0 && stamp_i < 64) az_stamps[wave_][stamp_i] = __builtin_amdgcn_s_memtime(); ++stamp_i; \
  } while (0)
#else
#define AZ_STAMP_AT() \
  do {                 \
  } while (0)
#endif

// Multiplies one unit out of the register slots W0/W1 and refills every slot, as soon
// as it has been consumed, with the same k-block of the wave's NEXT unit (same-slot
// refill: the prefetch distance is a whole unit, up to 32 KiB in flight per wave).
AZ_DEV void az_mlp_run_unit(const MlpUnit& u, const MlpUnit& next, az_f32x4 (&W0)[AZ_SLOTS],
                            az_f32x4 (&W1)[AZ_SLOTS], az_f32x4& acc0, az_f32x4& acc1, int lane,
                            int wave_, int& stamp_i) {
  if (u.first) {
    acc0 = az_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    acc1 = az_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // The biases are fetched BEFORE the refill loads are issued: vector loads return in
  // order, so a bias load issued after them would drain the whole prefetch.
  float bias0 = 0.0f, bias1 = 0.0f;
  if (u.last) {
    bias0 = u.bias[u.t0 * 16 + (lane & 15)];
    bias1 = u.two ? u.bias[(u.t0 + 1) * 16 + (lane & 15)] : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < AZ_SLOTS; ++kb) {
    if (kb < u.nkb) {
      const az_f32x4 a4 = *reinterpret_cast<const az_f32x4*>(u.ap + kb * 4);
      const az_f32x4 b0 = W0[kb], b1 = W1[kb];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
    }
    // keep the refill right behind its slot's MFMAs: left alone, the scheduler sinks
    // the loads to the end of the unit and the next unit then waits on fresh loads
    __builtin_amdgcn_sched_barrier(0);
    if (kb < next.nkb) {
      W0[kb] = next.w0[(size_t)kb * 64];
      W1[kb] = next.w1[(size_t)kb * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  AZ_STAMP_AT();
  if (!u.last) return;
  // epilogue.  C/D fragment: column = lane & 15, row = (lane >> 4) * 4 + r.
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1 && !u.two) break;
    const az_f32x4 acc = half ? acc1 : acc0;
    const int n = (u.t0 + half) * 16 + (lane & 15);
    const float bn = half ? bias1 : bias0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (lane >> 4) * 4 + r;
      float y = acc[r] + bn;
      if (u.out_act) {
        y = az_swish(y);
        u.out_act[row * u.out_stride + az_act_index(n, u.out_np)] = y;
      } else if (n < u.nout) {
        u.out_plain[row * u.out_stride + n] = y;
      }
    }
  }
}

// Both networks on the 16 rows staged in s.in0 -> s.logits / s.value.  Called by all
// NWAVES waves; ends with a workgroup barrier (LDS results visible to every thread).
template <int NWAVES>
AZ_DEV void az_forward_tile(const ModelDev& md, const MlpLds& s, int wave, int lane) {
  const int P = md.pol.n_layers > md.val.n_layers ? md.pol.n_layers : md.val.n_layers;
  az_f32x4 W0[AZ_SLOTS], W1[AZ_SLOTS];
  az_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int stamp_i = 0;
  const int wave_ = wave;
  (void)wave_;
  AZ_STAMP_AT();
  MlpUnit cur = az_mlp_next_unit<NWAVES>(md, s, P, 0, 0, 0, true, wave, lane);
#pragma unroll
  for (int kb = 0; kb < AZ_SLOTS; ++kb)
    if (kb < cur.nkb) {
      W0[kb] = cur.w0[(size_t)kb * 64];
      W1[kb] = cur.w1[(size_t)kb * 64];
    }
  for (int phase = 0; phase < P; ++phase) {
    while (cur.valid && cur.phase == phase) {
      const MlpUnit next =
          az_mlp_next_unit<NWAVES>(md, s, P, cur.phase, cur.idx, cur.round, false, wave, lane);
      AZ_STAMP_AT();
      az_mlp_run_unit(cur, next, W0, W1, acc0, acc1, lane, wave, stamp_i);
      AZ_STAMP_AT();
      cur = next;
    }
    AZ_STAMP_AT();
    az_lds_barrier();
    AZ_STAMP_AT();
  }
}

#endif  // AZ_MLP_H_
