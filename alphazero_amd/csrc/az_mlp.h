// az_mlp.h -- policy / value MLP forward for a tile of 16 batch lanes on the
// f32-input matrix cores (v_mfma_f32_16x16x4_f32).
//
// Why f32 MFMA: its result is bit for bit the k-ordered fused-multiply-add chain
//   acc = fma(x[k], W[k][n], acc),  k = 0, 1, 2, ...
// which is exactly how the CPU checker evaluates a dense layer, so the fused
// search stays bit-identical to the host evaluation while the contraction runs
// on the matrix pipe instead of the vector ALU.
//
// Data layout.  A dense layer in -> out is padded to Kp = ceil16(in), Np = ceil16(out).
//  * Packed weights: for output tile t (16 columns) and k-block kb (16 rows) one
//    float4 per lane,  Wp[t][kb][lane][q] = W[16*kb + 4*q + (lane>>4)][16*t + (lane&15)],
//    i.e. the B operand of four consecutive MFMA k-steps; a wave reads 1 KiB
//    contiguous per load.  Zero padded.  Packed bias: Np floats, zero padded.
//  * Activations in LDS: row m (lane of the tile) holds its Kp inputs permuted as
//    [k & 3][k >> 2] so that the A operands of four consecutive k-steps are one
//    aligned float4:  act[m*stride + (k&3)*(Kp/4) + (k>>2)].
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
  int max_kp;  // widest padded layer input/output over both networks
  MlpLayout pol, val;
  const float* packed;
  const float* mean;
  const float* std;
};

__host__ __device__ inline int az_ceil16(int x) { return (x + 15) & ~15; }

// Fills `L` from the layer sizes; returns the offset just past this network.
inline int64_t az_mlp_layout(int n_layers, const int32_t* sizes, int64_t base, MlpLayout* L) {
  L->n_layers = n_layers;
  int64_t off = base;
  for (int i = 0; i < n_layers; ++i) {
    L->kin[i] = sizes[i];
    L->nout[i] = sizes[i + 1];
    L->kp[i] = az_ceil16(sizes[i]);
    L->np[i] = az_ceil16(sizes[i + 1]);
    L->w_off[i] = off;
    off += (int64_t)L->kp[i] * L->np[i];
    L->b_off[i] = off;
    off += L->np[i];
  }
  return off;
}

// Activation row stride (floats) for a model whose widest padded layer is max_kp:
// +4 floats shifts consecutive rows by one 16-byte slot, which keeps the float4
// A-operand reads of the 16 rows on distinct LDS banks.
__host__ __device__ inline int az_act_stride(int max_kp) { return max_kp + 4; }

AZ_DEV int az_act_index(int k, int kp) { return (k & 3) * (kp >> 2) + (k >> 2); }

// One dense layer for a 16-row tile.  `in` / `out_act` are LDS activation buffers
// (permuted layout, row stride `stride`); for the last layer results go to
// `out_plain[m*out_stride + n]` instead.  Called by all NWAVES waves of the
// workgroup; the caller synchronises between layers.
template <int NWAVES>
AZ_DEV void az_dense_tile16(const MlpLayout& L, int layer, const float* __restrict__ packed,
                            const float* in, float* out_act, float* out_plain, int out_stride,
                            int stride, int wave, int lane) {
  const int kp = L.kp[layer], np = L.np[layer], nout = L.nout[layer];
  const int nkb = kp >> 4, ntiles = np >> 4;
  const bool last = layer == L.n_layers - 1;
  const az_f32x4* wbase = reinterpret_cast<const az_f32x4*>(packed + L.w_off[layer]);
  const float* bias = packed + L.b_off[layer];
  const int m = lane & 15, j = lane >> 4;
  const float* ap = in + m * stride + j * (kp >> 2);
  for (int t0 = wave * 2; t0 < ntiles; t0 += NWAVES * 2) {
    const bool two = t0 + 1 < ntiles;
    const az_f32x4* w0 = wbase + (size_t)t0 * nkb * 64 + lane;
    const az_f32x4* w1 = two ? w0 + (size_t)nkb * 64 : w0;
    az_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int kb = 0; kb < nkb; ++kb) {
      az_f32x4 a4 = *reinterpret_cast<const az_f32x4*>(ap + kb * 4);
      az_f32x4 b0 = w0[(size_t)kb * 64];
      az_f32x4 b1 = w1[(size_t)kb * 64];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
    }
    // C/D fragment: column = lane & 15, row = (lane >> 4) * 4 + r.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && !two) break;
      const az_f32x4 acc = half ? acc1 : acc0;
      const int n = (t0 + half) * 16 + (lane & 15);
      const float bn = bias[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float y = acc[r] + bn;
        if (!last) {
          y = az_swish(y);
          out_act[row * stride + az_act_index(n, np)] = y;
        } else if (n < nout) {
          out_plain[row * out_stride + n] = y;
        }
      }
    }
  }
}

// Whole network: input activations in `in0` (kept intact), ping-pong buffers
// bufa / bufb, result in out_plain[16][out_stride].  Ends with a barrier.
template <int NWAVES>
AZ_DEV void az_mlp_tile16(const MlpLayout& L, const float* __restrict__ packed, const float* in0,
                          float* bufa, float* bufb, float* out_plain, int out_stride, int stride,
                          int wave, int lane) {
  const float* in = in0;
  float* dst = bufa;
  for (int layer = 0; layer < L.n_layers; ++layer) {
    az_dense_tile16<NWAVES>(L, layer, packed, in, dst, out_plain, out_stride, stride, wave, lane);
    __syncthreads();
    in = dst;
    dst = (dst == bufa) ? bufb : bufa;
  }
}

// LDS floats needed by a workgroup that runs az_mlp_tile16 for both networks:
// in0 + two ping-pong buffers + logits [16][AZ_MAX_ACTIONS] + values [16][AZ_MAX_ATOMS].
__host__ __device__ inline int az_mlp_lds_floats(int max_kp) {
  return 3 * 16 * az_act_stride(max_kp) + 16 * AZ_MAX_ACTIONS + 16 * AZ_MAX_ATOMS;
}

#endif  // AZ_MLP_H_
