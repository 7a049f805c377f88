// az_mlp.h -- policy / value MLP forward for a tile of 16 batch lanes on the
// f32-input matrix cores (v_mfma_f32_16x16x4_f32).
//
// Why f32 MFMA: its result is bit for bit the k-ordered fused-multiply-add chain
//   acc = fma(x[k], W[k][n], acc),  k = 0, 1, 2, ...
// which is exactly how the CPU checker evaluates a dense layer, so the fused
// search stays bit-identical to the host evaluation while the contraction runs
// on the matrix pipe instead of the vector ALU.
//
// Data layout.  A dense layer in -> out is padded to Kp = ceil16(in) and Np = ceil16(out);
// the padding is zeros, which leave an accumulator that started at +0 unchanged.
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
// output tiles of the value layer; wave w takes items w, w + NWAVES, ...  A tile of a
// narrow policy layer (at most AZ_RIDER_SLOTS k-blocks) rides along on a value item
// as a third accumulator, so the small network adds a few MFMAs to a wave instead
// of a unit of its own on the phase's critical path; policy tiles that do not fit
// that way form pairs of their own after the value items.  With M = 16 every weight
// is used for only 16 multiply-adds, so a wave holds the 16 k-blocks of its current
// unit in 128 VGPRs and refills each register slot, right after the MFMAs that
// consumed it, with the same k-block of its NEXT unit -- which may belong to the next
// phase or, in the persistent search, to the next simulation, since weights do not
// depend on activations -- so up to 32 KiB per wave stay in flight behind the MFMAs,
// the epilogue, the barrier and the tree phase.  The unit descriptors of a persistent
// kernel are computed once into an LDS table.
#ifndef AZ_MLP_H_
#define AZ_MLP_H_

#include "az_device.h"
#include "az_swish.h"

typedef __attribute__((ext_vector_type(4))) float az_f32x4;

// Loads through pointers that are known to address global memory.  Pointers that
// went through the LDS descriptor table are generic to the compiler, and a generic
// (flat) load counts against the LDS counter as well, which would make every wait
// for an A operand wait for the weight prefetch too.
AZ_DEV az_f32x4 az_ldg4(const az_f32x4* p) {
  return *(const __attribute__((address_space(1))) az_f32x4*)p;
}
// element `lane` of a wave-uniform base: scalar base + 32-bit vector offset addressing
AZ_DEV az_f32x4 az_ldg4_lane(const az_f32x4* base, int lane) {
  return ((const __attribute__((address_space(1))) az_f32x4*)base)[(unsigned)lane];
}
AZ_DEV float az_ldg(const float* p) { return *(const __attribute__((address_space(1))) float*)p; }


struct MlpLayout {
  int n_layers;
  int kin[AZ_MAX_LAYERS], nout[AZ_MAX_LAYERS];
  int kp[AZ_MAX_LAYERS], np[AZ_MAX_LAYERS];
  int wb[AZ_MAX_LAYERS];  // 1: the layer's weights are packed as bfloat16 (az_mlp_layout)
  int64_t w_off[AZ_MAX_LAYERS], b_off[AZ_MAX_LAYERS];  // float offsets into the packed buffer
};

struct ModelDev {
  int env_id;
  float gamma;
  int obs_dim, A, K;
  int pol_kp, val_kp;  // widest padded layer of each network
  int bf16;            // 1: hidden layers on the bfloat16 matrix instructions (az_mlp_layout)
  int tab_units;       // units of the busiest wave in one pass over both networks (table rows per wave)
  int head_floats;     // LDS floats of the output heads' staged weights (az_mlp_head_floats), 0: not used
  MlpLayout pol, val;
  const float* packed;
  const float* mean;
  const float* std;
  const uint32_t* table;  // the schedule table az_model_pack left behind the weights (az_mlp_table_load)
  uint32_t table_hash;    // az_mlp_table_hash of this model: what that table's header must say
};

__host__ __device__ inline int az_ceil16(int x) { return (x + 15) & ~15; }
#define AZ_IN0_STRIDE 68  // row stride of the first layer's input buffer (Kp <= 64, + 4)
#define AZ_SLOTS 16       // k-blocks of one unit held in registers per tile
#define AZ_SLOTS_BF16 8   // ... in the kernels of a model with bfloat16 hidden layers (8 x 32 = 256 inputs)
#define AZ_RIDER_SLOTS 4  // k-blocks of a rider tile
#define AZ_MLP_WAVES 8    // waves of a workgroup that run the networks
#define AZ_TABLE_UNITS 16 // unit descriptors per wave in the LDS table
#define AZ_UNIT_WORDS 36  // 32-bit words of one descriptor (sizeof(MlpUnit) / 4)
// k-blocks per round of a unit.  With bfloat16 hidden layers a k-block of the 256-wide layers is
// 32 rows, and the weight stream from L2 -- not the matrix pipe -- bounds a phase: every one of a
// unit's slots is refilled whether it holds anything or not, so those kernels run 8 slots.
__host__ __device__ inline int az_mlp_slots(const ModelDev& md) { return md.bf16 ? AZ_SLOTS_BF16 : AZ_SLOTS; }

// Fills `L` from the layer sizes; returns the offset just past this network.
// hidden_bf16: the layers between the first and the last one keep their weights as bfloat16 and
// run on v_mfma_f32_16x16x32_bf16 (inputs rounded to bfloat16 as they are read, float32
// accumulation, float32 bias and activation function); the first layer -- the observation --
// and the output layer stay in float32.  One k-block of such a layer is 32 rows and takes the
// same 1 KiB per output tile as a float32 k-block of 16 rows: lane (n = lane & 15, g = lane >> 4)
// holds W[32 kb + 8 g + j][16 t + n], j = 0 .. 7.  The host checks that these layers' input widths
// are multiples of 32 (az_model_dev).
inline int64_t az_mlp_layout(int n_layers, const int32_t* sizes, int64_t base, MlpLayout* L, int hidden_bf16 = 0) {
  L->n_layers = n_layers;
  int64_t off = base;
  for (int i = 0; i < n_layers; ++i) {
    L->kin[i] = sizes[i];
    L->nout[i] = sizes[i + 1];
    L->kp[i] = az_ceil16(sizes[i]);
    L->np[i] = az_ceil16(sizes[i + 1]);
    L->wb[i] = hidden_bf16 && i >= 1 && i <= n_layers - 2;
    L->w_off[i] = off;
    off += L->wb[i] ? (int64_t)L->kp[i] * L->np[i] / 2 : (int64_t)L->kp[i] * L->np[i];
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
// per network, logits [16][AZ_MAX_ACTIONS], values [16][AZ_MAX_ATOMS], the unit table
// [AZ_MLP_WAVES][tab_units][AZ_UNIT_WORDS] (tab_units = the units of the busiest wave of this
// model, at most AZ_TABLE_UNITS), its per-wave counts, the observation normaliser's mean
// and standard deviation, and the staged weights of small output heads (az_mlp_head_floats).
__host__ __device__ inline int az_mlp_lds_floats(int pol_kp, int val_kp, int tab_units, int head_floats) {
  return 16 * AZ_IN0_STRIDE + 2 * 16 * az_act_stride(pol_kp) + 2 * 16 * az_act_stride(val_kp) +
         16 * AZ_MAX_ACTIONS + 16 * AZ_MAX_ATOMS +
         AZ_MLP_WAVES * tab_units * AZ_UNIT_WORDS + 16 + 2 * AZ_IN0_STRIDE + head_floats;
}

struct MlpLds {
  float *base, *in0, *pa, *pb, *va, *vb, *logits, *value;
  uint32_t* tab;   // unit table; 16-byte aligned because every region is a multiple of 4 floats
  int32_t* tab_n;  // units of each wave in the table, or -1 when they do not fit
  float *norm_mean, *norm_std;  // [obs_dim] copies of the observation normaliser
  float* head_w;                // staged output heads (az_mlp_heads)
  int pstride, vstride;
  int tab_units;  // rows of the table per wave
  AZ_DEV MlpLds(float* base_, int pol_kp, int val_kp, int tab_units_) {
    tab_units = tab_units_;
    base = base_;
    pstride = az_act_stride(pol_kp);
    vstride = az_act_stride(val_kp);
    in0 = base;
    pa = in0 + 16 * AZ_IN0_STRIDE;
    pb = pa + 16 * pstride;
    va = pb + 16 * pstride;
    vb = va + 16 * vstride;
    logits = vb + 16 * vstride;
    value = logits + 16 * AZ_MAX_ACTIONS;
    tab = reinterpret_cast<uint32_t*>(value + 16 * AZ_MAX_ATOMS);
    tab_n = reinterpret_cast<int32_t*>(tab + AZ_MLP_WAVES * tab_units * AZ_UNIT_WORDS);
    norm_mean = reinterpret_cast<float*>(tab_n + 16);
    norm_std = norm_mean + AZ_IN0_STRIDE;
    head_w = norm_std + AZ_IN0_STRIDE;
  }
};

// Where the results of one 16-column output tile go.
struct MlpTile {
  const float* bias;  // at the tile's first column
  int out_off;        // float offset from MlpLds::base of the output buffer
  int out_stride, out_np, nout, n0;
  int act;            // 1: swish into the next layer's permuted activations; 0: plain [16][out_stride]
};

// AZ_U_BF16 / AZ_U_RIDER_BF16: the pair's / the rider's layer is a bfloat16 one (k-blocks of 32)
enum { AZ_U_VALID = 1, AZ_U_TWO = 2, AZ_U_FIRST = 4, AZ_U_LAST = 8, AZ_U_RIDER = 16, AZ_U_BF16 = 32,
       AZ_U_RIDER_BF16 = 64 };

// One unit of work: a pair of output tiles of one layer, restricted to one round of
// AZ_SLOTS k-blocks (layers wider than 16*AZ_SLOTS inputs take several rounds; the
// accumulators carry over), plus an optional rider tile of the other network.
// Plain data without lane-dependent parts, so that it can live in the LDS table.
struct MlpUnit {
  const az_f32x4 *w0, *w1, *w2;  // packed weights of tiles t0, t0 + 1 and the rider (kb stride 64)
  MlpTile o0, o2;                // outputs of the pair (second tile: columns n0 + 16 ..) and the rider
  int ap_off, in_stride, kpq, nkb;      // A operands of the pair: LDS offset, row stride, Kp / 4
  int ap2_off, in_stride2, kpq2, nkb2;  // A operands of the rider
  int phase, idx, round, flags;
  int bars;  // phase barriers the wave passes after this unit (0: its next unit is in the same phase)
  int pad;
};

// What the refills need to know about the unit AFTER the running one.
struct MlpPre {
  const az_f32x4 *w0, *w1, *w2;
  int nkb, nkb2;
};
static_assert(sizeof(MlpPre) == 8 * 4, "MlpPre is read as 8 words");
static_assert(sizeof(MlpUnit) == AZ_UNIT_WORDS * 4, "AZ_UNIT_WORDS must match MlpUnit");

struct PhaseShape {
  int n_val;     // pairs of value tiles
  int n_riders;  // policy tiles riding on value items 0 .. n_riders - 1
  int n_pol;     // pairs of the remaining policy tiles
  int n;         // items of the phase
};

__host__ __device__ inline PhaseShape az_phase_shape(const ModelDev& md, int phase) {
  PhaseShape p;
  const int tv = phase < md.val.n_layers ? md.val.np[phase] >> 4 : 0;
  const int tp = phase < md.pol.n_layers ? md.pol.np[phase] >> 4 : 0;
  p.n_val = (tv + 1) >> 1;
  const bool ride = p.n_val > 0 && tp > 0 && (md.pol.kp[phase] >> 4) <= AZ_RIDER_SLOTS;
  p.n_riders = ride ? (tp < p.n_val ? tp : p.n_val) : 0;
  p.n_pol = (tp - p.n_riders + 1) >> 1;
  p.n = p.n_val + p.n_pol;
  return p;
}

AZ_DEV void az_mlp_tile_in(const ModelDev& md, const MlpLds& s, bool is_val, int phase, int kb_base,
                           int& ap_off, int& in_stride, int& kpq) {
  const MlpLayout& L = is_val ? md.val : md.pol;
  const int kb_quads = L.wb[phase] ? 8 : 4;  // floats of one k-block in each of the four [k & 3] planes
  const float* a = is_val ? s.va : s.pa;
  const float* b = is_val ? s.vb : s.pb;
  // phase 0 reads in0; phase i > 0 reads the buffer phase i-1 wrote
  const float* in = phase == 0 ? s.in0 : ((phase & 1) ? a : b);
  in_stride = phase == 0 ? AZ_IN0_STRIDE : (is_val ? s.vstride : s.pstride);
  kpq = L.kp[phase] >> 2;
  ap_off = (int)(in - s.base) + kb_base * kb_quads;
}

AZ_DEV void az_mlp_tile_out(const ModelDev& md, const MlpLds& s, bool is_val, int phase, int tile,
                            MlpTile& o) {
  const MlpLayout& L = is_val ? md.val : md.pol;
  const bool last_layer = phase == L.n_layers - 1;
  float* a = is_val ? s.va : s.pa;
  float* b = is_val ? s.vb : s.pb;
  float* out = last_layer ? (is_val ? s.value : s.logits) : ((phase & 1) ? b : a);
  o.bias = md.packed + L.b_off[phase] + tile * 16;
  o.out_off = (int)(out - s.base);
  o.out_stride = last_layer ? (is_val ? AZ_MAX_ATOMS : AZ_MAX_ACTIONS) : (is_val ? s.vstride : s.pstride);
  o.out_np = L.np[phase];
  o.nout = L.nout[phase];
  o.n0 = tile * 16;
  o.act = last_layer ? 0 : 1;
}

AZ_DEV MlpUnit az_mlp_unit(const ModelDev& md, const MlpLds& s, int phase, int idx, int round) {
  MlpUnit u;
  u.phase = phase;
  u.idx = idx;
  u.round = round;
  u.bars = 0;
  u.pad = 0;
  const PhaseShape ps = az_phase_shape(md, phase);
  const bool is_val = idx < ps.n_val;
  const MlpLayout& L = is_val ? md.val : md.pol;
  const int t0 = is_val ? idx * 2 : ps.n_riders + (idx - ps.n_val) * 2;
  const int nkb_total = L.kp[phase] >> (L.wb[phase] ? 5 : 4);
  const int slots = az_mlp_slots(md);
  const int kb_base = round * slots;
  u.nkb = nkb_total - kb_base < slots ? nkb_total - kb_base : slots;
  const bool two = t0 + 1 < (L.np[phase] >> 4);
  const bool rider = is_val && idx < ps.n_riders && round == 0;
  u.flags = AZ_U_VALID | (two ? AZ_U_TWO : 0) | (round == 0 ? AZ_U_FIRST : 0) |
            (kb_base + slots >= nkb_total ? AZ_U_LAST : 0) | (rider ? AZ_U_RIDER : 0) |
            (L.wb[phase] ? AZ_U_BF16 : 0) | (rider && md.pol.wb[phase] ? AZ_U_RIDER_BF16 : 0);
  const az_f32x4* wbase = reinterpret_cast<const az_f32x4*>(md.packed + L.w_off[phase]);
  u.w0 = wbase + ((size_t)t0 * nkb_total + kb_base) * 64;
  u.w1 = two ? u.w0 + (size_t)nkb_total * 64 : u.w0;
  az_mlp_tile_in(md, s, is_val, phase, kb_base, u.ap_off, u.in_stride, u.kpq);
  az_mlp_tile_out(md, s, is_val, phase, t0, u.o0);
  if (rider) {
    const int nkb2 = md.pol.kp[phase] >> (md.pol.wb[phase] ? 5 : 4);
    u.w2 = reinterpret_cast<const az_f32x4*>(md.packed + md.pol.w_off[phase]) + (size_t)idx * nkb2 * 64;
    u.nkb2 = nkb2;
    az_mlp_tile_in(md, s, false, phase, 0, u.ap2_off, u.in_stride2, u.kpq2);
    az_mlp_tile_out(md, s, false, phase, idx, u.o2);
  } else {
    u.w2 = u.w0;
    u.nkb2 = 0;
    u.ap2_off = u.ap_off;
    u.in_stride2 = u.in_stride;
    u.kpq2 = u.kpq;
    u.o2 = u.o0;
  }
  return u;
}

AZ_DEV MlpUnit az_mlp_no_unit(int P) {
  MlpUnit none{};
  none.flags = 0;
  none.phase = P;
  none.nkb = 0;
  none.nkb2 = 0;
  return none;
}

// The unit wave `wave` runs after (phase, idx, round); `start` asks for its first unit.
template <int NWAVES>
AZ_DEV MlpUnit az_mlp_next_unit(const ModelDev& md, const MlpLds& s, int P, int phase, int idx,
                                int round, bool start, int wave) {
  if (!start) {
    const PhaseShape ps = az_phase_shape(md, phase);
    const MlpLayout& L = idx < ps.n_val ? md.val : md.pol;
    const int nrounds = ((L.kp[phase] >> (L.wb[phase] ? 5 : 4)) + az_mlp_slots(md) - 1) / az_mlp_slots(md);
    if (round + 1 < nrounds) return az_mlp_unit(md, s, phase, idx, round + 1);
    if (idx + NWAVES < ps.n) return az_mlp_unit(md, s, phase, idx + NWAVES, 0);
    phase += 1;
  } else {
    phase = 0;
  }
  for (; phase < P; ++phase)
    if (wave < az_phase_shape(md, phase).n) return az_mlp_unit(md, s, phase, wave, 0);
  return az_mlp_no_unit(P);
}

__host__ __device__ inline int az_mlp_phases(const ModelDev& md) {
  return md.pol.n_layers > md.val.n_layers ? md.pol.n_layers : md.val.n_layers;
}

// LDS floats for the output heads when the last phase consists of small output heads only (at
// most AZ_MLP_WAVES output columns between them; a network that ends earlier has nothing in
// that phase), else 0.  Such a phase is a handful of matrix-vector products: as a unit of the
// matrix pipe it would occupy one wave (which multiplies its single tile twice, the loop always
// running two accumulators) while seven wait at the last barrier; it runs on the vector ALU
// instead, one wave per output column (az_mlp_heads), from weights staged once per kernel in
// LDS, and the unit machinery covers the phases before it.  Per head: nout columns of kp
// weights + 16 biases.
// Which layers these are is part of the numerics specification (az_small_heads, az_math.h: they
// are summed as four interleaved chains); this function only sizes their LDS copy.
__host__ __device__ inline int az_mlp_head_floats(const ModelDev& md) {
  const int P = az_mlp_phases(md);
  const int lp = md.pol.n_layers, lv = md.val.n_layers;
  const int mask = az_small_heads(lp, lp > 0 ? md.pol.kin[lp - 1] : 0, lp > 0 ? md.pol.nout[lp - 1] : 0,
                                  lv, lv > 0 ? md.val.kin[lv - 1] : 0, lv > 0 ? md.val.nout[lv - 1] : 0);
  int total = 0;
  if (mask & 1) total += md.pol.nout[P - 1] * md.pol.kp[P - 1] + 16;
  if (mask & 2) total += md.val.nout[P - 1] * md.val.kp[P - 1] + 16;
  static_assert(AZ_MLP_WAVES == 8, "az_small_heads allows one output column per wave");
  return total;
}
__host__ __device__ inline bool az_mlp_heads_apply(const ModelDev& md) { return md.head_floats > 0; }
// phases that run as units of the matrix pipe
__host__ __device__ inline int az_mlp_unit_phases(const ModelDev& md) {
  return az_mlp_phases(md) - (az_mlp_heads_apply(md) ? 1 : 0);
}

// Units the busiest wave runs in one pass over both networks (a wave's items are idx = wave,
// wave + AZ_MLP_WAVES, ...; value and policy items of a phase can take different numbers of
// rounds, so the busiest wave is not always wave 0); the schedule table holds that many per wave
// (ModelDev::tab_units); the host refuses models that need more than AZ_TABLE_UNITS.
inline int az_mlp_units_per_wave(const ModelDev& md) {
  int most = 0;
  for (int wave = 0; wave < AZ_MLP_WAVES; ++wave) {
    int units = 0;
    for (int phase = 0; phase < az_mlp_unit_phases(md); ++phase) {
      const PhaseShape ps = az_phase_shape(md, phase);
      for (int idx = wave; idx < ps.n; idx += AZ_MLP_WAVES) {
        const MlpLayout& L = idx < ps.n_val ? md.val : md.pol;
        units += ((L.kp[phase] >> (L.wb[phase] ? 5 : 4)) + az_mlp_slots(md) - 1) / az_mlp_slots(md);
      }
    }
    most = units > most ? units : most;
  }
  return most;
}

// Writes the wave's unit sequence into the LDS table (every wave calls it; the caller
// places a workgroup barrier before the first az_forward_tile).  tab_n[wave] = number of
// units, tab_n[NWAVES + wave] = phase barriers the wave passes before its first unit.
template <int NWAVES>
AZ_DEV void az_mlp_table_build(const ModelDev& md, const MlpLds& s, int wave, int lane) {
  const int P = az_mlp_unit_phases(md);  // (the barrier counts below end where the heads begin)
  int n = 0;
  MlpUnit u = az_mlp_next_unit<NWAVES>(md, s, P, 0, 0, 0, true, wave);
  const int lead = u.phase;  // P when the wave has no unit at all
  while (u.flags & AZ_U_VALID) {
    const MlpUnit nx = az_mlp_next_unit<NWAVES>(md, s, P, u.phase, u.idx, u.round, false, wave);
    u.bars = nx.phase - u.phase;  // past the last unit: nx.phase == P
    if (n < s.tab_units && lane == 0)
      *reinterpret_cast<MlpUnit*>(s.tab + (size_t)(wave * s.tab_units + n) * AZ_UNIT_WORDS) = u;
    ++n;
    u = nx;
  }
  if (lane == 0) {
    s.tab_n[wave] = n <= s.tab_units ? n : 0;
    s.tab_n[NWAVES + wave] = n <= s.tab_units ? lead : P;
  }
}

// The schedule table behind the packed weights.  Building it takes a wave ~25 k cycles of scalar
// code (ten microseconds per launch of every kernel that runs the networks), and it depends on
// the layer shapes only, so az_model_pack builds it once, with these same functions
// (k_table_build), and leaves it behind the weights:
//   word 0 az_mlp_table_hash of the model it was built for, word 1 its rows per wave (tab_units),
//   words 2-3 the address `packed` had at that time, words 4-19 MlpLds::tab_n, then
//   MlpLds::tab ([AZ_MLP_WAVES][tab_units][AZ_UNIT_WORDS]).
// A kernel whose model matches the header copies it into LDS (az_mlp_table_load), moving the
// pointers inside by the distance the buffer has moved since; any other buffer -- packed for other
// shapes, or not by az_model_pack -- makes the kernel build the table itself as before.
#define AZ_TABLE_HDR_WORDS 20
#define AZ_TABLE_MAX_WORDS (AZ_TABLE_HDR_WORDS + AZ_MLP_WAVES * AZ_TABLE_UNITS * AZ_UNIT_WORDS)
static_assert(AZ_TABLE_MAX_WORDS % 4 == 0 && AZ_UNIT_WORDS % 2 == 0, "the table is copied in 8-byte pairs");
static_assert(offsetof(MlpUnit, w0) == 0 && offsetof(MlpUnit, w1) == 8 && offsetof(MlpUnit, w2) == 16 &&
                  offsetof(MlpUnit, o0) + offsetof(MlpTile, bias) == 24 &&
                  offsetof(MlpUnit, o2) + offsetof(MlpTile, bias) == 56,
              "pairs 0, 1, 2, 3 and 7 of a unit are its pointers (az_mlp_table_load)");

__host__ __device__ inline uint32_t az_mlp_table_hash(const ModelDev& md) {
  uint32_t h = 2166136261u;
  auto mix = [&](int v) { h = (h ^ (uint32_t)v) * 16777619u; };
  for (int net = 0; net < 2; ++net) {
    const MlpLayout& L = net ? md.val : md.pol;
    mix(L.n_layers);
    for (int i = 0; i < L.n_layers; ++i) {
      mix(L.kin[i]);
      mix(L.nout[i]);
      mix(L.wb[i]);
      mix((int)L.w_off[i]);
      mix((int)L.b_off[i]);
    }
  }
  mix(md.bf16);
  mix(md.tab_units);
  mix(md.head_floats);
  mix(AZ_UNIT_WORDS * 1000 + AZ_TABLE_UNITS);
  return h | 1u;  // never 0: a zeroed header matches nothing
}

// Workgroup-wide; true when the table was taken from the packed buffer (the caller's barrier
// publishes it, as it does a table built in place).
AZ_DEV bool az_mlp_table_load(const ModelDev& md, const MlpLds& s, int tid, int nthreads) {
  const uint32_t* g = md.table;
  if (g == nullptr) return false;
  const uint32_t h0 = __builtin_amdgcn_readfirstlane(g[0]), h1 = __builtin_amdgcn_readfirstlane(g[1]);
  if (h0 != md.table_hash || (int)h1 != md.tab_units) return false;
  const uint64_t* g2 = reinterpret_cast<const uint64_t*>(g);
  const uint64_t delta = (uint64_t)(uintptr_t)md.packed - g2[1];
  for (int i = tid; i < 16; i += nthreads) s.tab_n[i] = (int32_t)g[4 + i];
  const int pairs = AZ_MLP_WAVES * md.tab_units * (AZ_UNIT_WORDS / 2);
  uint64_t* dst = reinterpret_cast<uint64_t*>(s.tab);
  for (int p = tid; p < pairs; p += nthreads) {
    const int q = p % (AZ_UNIT_WORDS / 2);
    uint64_t v = g2[AZ_TABLE_HDR_WORDS / 2 + p];
    if (q < 4 || q == 7) v += delta;
    dst[p] = v;
  }
  return true;
}

template <int NWAVES>
AZ_DEV void az_mlp_table_setup(const ModelDev& md, const MlpLds& s, int tid, int nthreads, int wave, int lane) {
  if (!az_mlp_table_load(md, s, tid, nthreads)) az_mlp_table_build<NWAVES>(md, s, wave, lane);
}

// One LDS access for everything a wave needs at the start of unit `i`: lane k < AZ_UNIT_WORDS
// reads word k of descriptor i, lanes 40 .. 47 the refill fields (MlpPre) of descriptor `ipre`.
#define AZ_PRE_LANE0 40
AZ_DEV uint32_t az_mlp_table_read(const MlpLds& s, int wave, int i, int ipre, int lane) {
  static_assert(AZ_UNIT_WORDS <= AZ_PRE_LANE0 && AZ_PRE_LANE0 + 8 <= 64, "one word per lane");
  const int pw = lane - AZ_PRE_LANE0;  // 0..5: the three weight pointers, 6: nkb, 7: nkb2
  const int pre_word = pw < 6 ? pw : (pw == 6 ? (int)(offsetof(MlpUnit, nkb) / 4) : (int)(offsetof(MlpUnit, nkb2) / 4));
  int word = 0;
  if (lane < AZ_UNIT_WORDS) word = i * AZ_UNIT_WORDS + lane;
  if (pw >= 0 && pw < 8) word = ipre * AZ_UNIT_WORDS + pre_word;
  return s.tab[wave * (s.tab_units * AZ_UNIT_WORDS) + word];
}

// The words broadcast into scalar registers.  The weight pointers of the running unit are
// not needed (its weights are in registers already) and are left zero.
AZ_DEV void az_mlp_table_decode(uint32_t mine, MlpUnit& u, MlpPre& pre) {
  union {
    MlpUnit u;
    uint32_t w[AZ_UNIT_WORDS];
  } x;
#pragma unroll
  for (int k = 0; k < AZ_UNIT_WORDS; ++k) x.w[k] = k < 6 ? 0u : (uint32_t)__builtin_amdgcn_readlane(mine, k);
  u = x.u;
  union {
    MlpPre p;
    uint32_t w[8];
  } y;
#pragma unroll
  for (int k = 0; k < 8; ++k) y.w[k] = __builtin_amdgcn_readlane(mine, AZ_PRE_LANE0 + k);
  pre = y.p;
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

// Register state of a wave between units: the weight slots.
struct MlpRegs {
  az_f32x4 W0[AZ_SLOTS], W1[AZ_SLOTS], W2[AZ_RIDER_SLOTS];
};

// One output tile's epilogue.  C/D fragment: column = lane & 15, row = (lane >> 4) * 4 + r.
AZ_DEV void az_mlp_tile_epilogue(const MlpTile& o, int n_off, const az_f32x4& acc, float bn,
                                 float* lds_base, int lane) {
  const int n = o.n0 + n_off + (lane & 15);
  float* out = lds_base + o.out_off + (lane >> 4) * 4 * o.out_stride;
  if (o.act) {
    az_f32x2 v[2] = {az_f32x2{acc.x, acc.y} + az_splat2(bn), az_f32x2{acc.z, acc.w} + az_splat2(bn)};
    az_swish2xn<2>(v);
    const az_f32x2 lo = v[0], hi = v[1];
    float* q = out + az_act_index(n, o.out_np);
    q[0] = lo.x;
    q[o.out_stride] = lo.y;
    q[2 * o.out_stride] = hi.x;
    q[3 * o.out_stride] = hi.y;
  } else if (n < o.nout) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r * o.out_stride + n] = acc[r] + bn;
  }
}

// Multiplies one unit out of the register slots and refills every slot, as soon as it
// has been consumed, with the same k-block of the wave's NEXT unit (same-slot refill:
// the prefetch distance is a whole unit, up to 32 KiB in flight per wave).  There is
// one copy of this loop on purpose: a second instantiation (say, for single-tile
// units) makes the register allocator keep two sets of weight slots and spill.
// The A operand of one bfloat16 k-block (32 inputs) for this lane: inputs 32 kb + 8 g + j, j = 0 .. 7
// (g = lane >> 4), out of the float32 activations in their [k & 3][k >> 2] order -- two
// consecutive floats out of each of the four planes -- rounded to bfloat16 (nearest even).
// `apb` = the lane's row + 2 g, `kpq` = floats per plane.
typedef __attribute__((ext_vector_type(8))) __bf16 az_bf16x8;
typedef __attribute__((ext_vector_type(2))) float az_f32x2_;
struct MlpABf16 {
  az_f32x2_ p[4];
};
AZ_DEV MlpABf16 az_mlp_a_bf16_read(const float* apb, int kpq, int kb) {
  MlpABf16 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.p[j] = *reinterpret_cast<const az_f32x2_*>(apb + j * kpq + kb * 8);
  return r;
}
AZ_DEV az_bf16x8 az_mlp_a_bf16(const MlpABf16& r) {
  az_bf16x8 a;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = (__bf16)r.p[j].x;      // k & 3 = j, first of the two
    a[4 + j] = (__bf16)r.p[j].y;  // the next four inputs
  }
  return a;
}
AZ_DEV az_bf16x8 az_as_bf16x8(const az_f32x4& w) { return __builtin_bit_cast(az_bf16x8, w); }

// BF16MODE: the kernel belongs to a model with bfloat16 hidden layers; a unit's flags say which of
// its parts are (in the other kernels that code does not exist).
template <bool BF16MODE>
AZ_DEV void az_mlp_unit_loop(const MlpUnit& u, const MlpPre& next, MlpRegs& R, az_f32x4& acc0,
                             az_f32x4& acc1, az_f32x4& acc2, const float* ap, const float* ap2,
                             int lane) {
  const int nk = next.nkb - 1, nk2 = next.nkb2 > 0 ? next.nkb2 - 1 : 0;
  az_f32x4 a[2];  // A operands of even / odd k-blocks
  const bool pair_bf16 = BF16MODE && (u.flags & AZ_U_BF16), rider_bf16 = BF16MODE && (u.flags & AZ_U_RIDER_BF16);
  // (bfloat16 parts: the same row, counted from input 8 g instead of plane g)
  const int g = lane >> 4;
  const float* apb = ap - g * u.kpq + 2 * g;
  const float* apb2 = ap2 - g * u.kpq2 + 2 * g;
  MlpABf16 ab[2];
  if (pair_bf16)
    ab[0] = az_mlp_a_bf16_read(apb, u.kpq, 0);
  else
    a[0] = *reinterpret_cast<const az_f32x4*>(ap);
  constexpr int kSlots = BF16MODE ? AZ_SLOTS_BF16 : AZ_SLOTS;
#pragma unroll
  for (int kb = 0; kb < kSlots; ++kb) {
    if (BF16MODE && pair_bf16) {
      if (kb < u.nkb) {
        if (kb + 1 < kSlots) ab[(kb + 1) & 1] = az_mlp_a_bf16_read(apb, u.kpq, kb + 1);
        const az_bf16x8 a8 = az_mlp_a_bf16(ab[kb & 1]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, az_as_bf16x8(R.W0[kb]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, az_as_bf16x8(R.W1[kb]), acc1, 0, 0, 0);
      }
    } else
    if (kb < u.nkb) {
      // The A operands of the next k-block are read while this one multiplies --
      // unconditionally, so that the wait below can leave exactly this one read in
      // flight; past the unit's last k-block it reads LDS that is not used.
      if (kb + 1 < kSlots) a[(kb + 1) & 1] = *reinterpret_cast<const az_f32x4*>(ap + (kb + 1) * 4);
      const az_f32x4 a4 = a[kb & 1];
      const az_f32x4 b0 = R.W0[kb];
      // a single-tile unit multiplies its tile twice (w1 == w0; acc1 is ignored): a
      // branch here would cost accumulator copies that drain the MFMA pipe per k-block
      const az_f32x4 b1 = R.W1[kb];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
    }
    if (BF16MODE && rider_bf16) {
      if (kb < AZ_RIDER_SLOTS && kb < u.nkb2) {
        const az_bf16x8 a8 = az_mlp_a_bf16(az_mlp_a_bf16_read(apb2, u.kpq2, kb));
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, az_as_bf16x8(R.W2[kb < AZ_RIDER_SLOTS ? kb : 0]), acc2, 0, 0, 0);
      }
    } else
    if (kb < AZ_RIDER_SLOTS && kb < u.nkb2) {
      const az_f32x4 a2 = *reinterpret_cast<const az_f32x4*>(ap2 + kb * 4);
      const az_f32x4 b2 = R.W2[kb < AZ_RIDER_SLOTS ? kb : 0];
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, b2.z, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, b2.w, acc2, 0, 0, 0);
    }
    // keep the refill right behind its slot's MFMAs: left alone, the scheduler sinks
    // the loads to the end of the unit and the next unit then waits on fresh loads.
    // Every slot is refilled unconditionally (a unit with fewer k-blocks re-reads its
    // last one): with a fixed number of loads per unit the compiler can wait for a slot
    // with the exact distance in the in-order load counter, whereas a conditional load
    // forces it to drain the whole prefetch (vmcnt(0)) before every k-block.
    __builtin_amdgcn_sched_barrier(0);
    {
      const int kc = kb < nk ? kb : nk;
      R.W0[kb] = az_ldg4_lane(next.w0 + (size_t)kc * 64, lane);
      R.W1[kb] = az_ldg4_lane(next.w1 + (size_t)kc * 64, lane);
      if (kb < AZ_RIDER_SLOTS) {
        const int kc2 = kb < nk2 ? kb : nk2;
        R.W2[kb < AZ_RIDER_SLOTS ? kb : 0] = az_ldg4_lane(next.w2 + (size_t)kc2 * 64, lane);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// What a unit does with its accumulators: the rider's tile, and -- after the last round of
// k-blocks -- the pair's tiles.
AZ_DEV void az_mlp_unit_epilogue(const MlpUnit& u, const az_f32x4& acc0, const az_f32x4& acc1,
                                 const az_f32x4& acc2, float bias0, float bias1, float bias2,
                                 float* lds_base, int lane) {
  const bool two = u.flags & AZ_U_TWO, last = u.flags & AZ_U_LAST, rider = u.flags & AZ_U_RIDER;
  if (rider) az_mlp_tile_epilogue(u.o2, 0, acc2, bias2, lds_base, lane);
  if (!last) return;
  if (u.o0.act) {
    // Both tiles' activations in one basic block: four independent chains of packed
    // instructions, interleaved (az_swish2x4), where one tile after the other runs at the
    // latency of a single chain.  (A single-tile unit has acc1 == acc0; its second
    // tile is computed and dropped.)
    const MlpTile& o = u.o0;
    az_f32x2 v[4] = {az_f32x2{acc0.x, acc0.y} + az_splat2(bias0), az_f32x2{acc0.z, acc0.w} + az_splat2(bias0),
                     az_f32x2{acc1.x, acc1.y} + az_splat2(bias1), az_f32x2{acc1.z, acc1.w} + az_splat2(bias1)};
    az_swish2x4(v);
    const az_f32x2 v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    float* out = lds_base + o.out_off + (lane >> 4) * 4 * o.out_stride;
    float* q = out + az_act_index(o.n0 + (lane & 15), o.out_np);
    q[0] = v0.x;
    q[o.out_stride] = v0.y;
    q[2 * o.out_stride] = v1.x;
    q[3 * o.out_stride] = v1.y;
    if (two) {
      float* q1 = out + az_act_index(o.n0 + 16 + (lane & 15), o.out_np);
      q1[0] = v2.x;
      q1[o.out_stride] = v2.y;
      q1[2 * o.out_stride] = v3.x;
      q1[3 * o.out_stride] = v3.y;
    }
    return;
  }
  az_mlp_tile_epilogue(u.o0, 0, acc0, bias0, lds_base, lane);
  if (two) az_mlp_tile_epilogue(u.o0, 16, acc1, bias1, lds_base, lane);
}


template <bool BF16MODE>
AZ_DEV void az_mlp_run_unit(const MlpUnit& u, const MlpPre& next, MlpRegs& R, az_f32x4& acc0,
                            az_f32x4& acc1, float* lds_base, int lane, int wave_, int& stamp_i) {
  const bool two = u.flags & AZ_U_TWO;
  if (u.flags & AZ_U_FIRST) {
    acc0 = az_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    acc1 = az_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  az_f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
  // The biases are fetched BEFORE the refill loads are issued: vector loads return in
  // order, so a bias load issued after them would drain the whole prefetch.
  // All three are always loaded, for the same reason as the unconditional refills.
  const float bias0 = az_ldg(u.o0.bias + (lane & 15));
  const float bias1 = az_ldg(u.o0.bias + (two ? 16 : 0) + (lane & 15));
  const float bias2 = az_ldg(u.o2.bias + (lane & 15));
  const int lane_row = lane & 15, lane_q = lane >> 4;
  const float* ap = lds_base + u.ap_off + lane_row * u.in_stride + lane_q * u.kpq;
  const float* ap2 = lds_base + u.ap2_off + lane_row * u.in_stride2 + lane_q * u.kpq2;
  __builtin_amdgcn_sched_barrier(0);
  // Two waves share a SIMD.  The one that comes first keeps the matrix pipe (the arbiter
  // favours it: measured, its k-loop ends after ~5.8 k cycles, its partner's after ~9.4 k,
  // with or without different priorities for the two), and its epilogue then fills the issue
  // slots between its partner's matrix instructions.  The k-loop runs above the epilogue's
  // priority.
  __builtin_amdgcn_s_setprio(2);
  az_mlp_unit_loop<BF16MODE>(u, next, R, acc0, acc1, acc2, ap, ap2, lane);
  __builtin_amdgcn_s_setprio(0);
  AZ_STAMP_AT();
  az_mlp_unit_epilogue(u, acc0, acc1, acc2, bias0, bias1, bias2, lds_base, lane);
}

// The wave's first unit when it is a single k-block (every network whose input is at most 16
// wide starts so): its few weights and biases, fetched on their own.  Through the register slots
// such a unit would cost a whole set of AZ_PRIME_LOADS loads of its own and, worse, would fetch
// the NEXT unit's set -- a quarter of a megabyte per workgroup -- while it runs, with nothing to
// hide that behind; handled directly, the slots can be primed with the second unit's weights
// long before the networks start.
struct MlpFirst {
  az_f32x4 w0, w1, w2;
  float bias0, bias1, bias2;
  int direct;  // wave-uniform: the first unit runs from these registers, the slots hold the second unit
};

// Loads the weights of the wave's first unit; call before az_forward_tile (once, when
// the calls wrap).  In two steps for callers that spread the loads over other work:
// az_forward_prime_begin fetches the refill fields (and, with allow_direct, decides on the direct
// form and requests its operands first of all), az_forward_prime_slots issues the loads of the
// slots [KB0, KB1) -- every slot exactly once, in ascending order.
template <int NWAVES>
AZ_DEV MlpPre az_forward_prime_begin(const MlpLds& s, int wave, int lane, MlpFirst& F, bool allow_direct) {
  const int n_tab = __builtin_amdgcn_readfirstlane(s.tab_n[wave]);
  const int lead = __builtin_amdgcn_readfirstlane(s.tab_n[NWAVES + wave]);
  // lanes below AZ_UNIT_WORDS: the words of unit 0; lanes AZ_PRE_LANE0 ..: the refill fields of unit 1
  const uint32_t mine = az_mlp_table_read(s, wave, 0, n_tab > 1 ? 1 : 0, lane);
  auto word = [&](size_t byte_offset) { return (uint32_t)__builtin_amdgcn_readlane(mine, (int)(byte_offset / 4)); };
  const int flags = (int)word(offsetof(MlpUnit, flags));
  const int nkb = (int)word(offsetof(MlpUnit, nkb)), nkb2 = (int)word(offsetof(MlpUnit, nkb2));
  F.direct = allow_direct && n_tab > 1 && lead == 0 && nkb == 1 && nkb2 <= 1 && (flags & AZ_U_FIRST) &&
             (flags & AZ_U_LAST);
  MlpUnit unused;
  MlpPre cur;
  if (F.direct) {
    auto ptr = [&](size_t byte_offset) {
      const uint64_t lo = word(byte_offset), hi = word(byte_offset + 4);
      return (lo | (hi << 32));
    };
    const az_f32x4* w0 = reinterpret_cast<const az_f32x4*>(ptr(offsetof(MlpUnit, w0)));
    const az_f32x4* w1 = reinterpret_cast<const az_f32x4*>(ptr(offsetof(MlpUnit, w1)));
    const az_f32x4* w2 = reinterpret_cast<const az_f32x4*>(ptr(offsetof(MlpUnit, w2)));
    const float* b0 = reinterpret_cast<const float*>(ptr(offsetof(MlpUnit, o0) + offsetof(MlpTile, bias)));
    const float* b2 = reinterpret_cast<const float*>(ptr(offsetof(MlpUnit, o2) + offsetof(MlpTile, bias)));
    F.w0 = az_ldg4_lane(w0, lane);
    F.w1 = az_ldg4_lane(w1, lane);
    F.w2 = az_ldg4_lane(w2, lane);
    F.bias0 = az_ldg(b0 + (lane & 15));
    F.bias1 = az_ldg(b0 + ((flags & AZ_U_TWO) ? 16 : 0) + (lane & 15));
    F.bias2 = az_ldg(b2 + (lane & 15));
    az_mlp_table_decode(mine, unused, cur);  // (the refill fields of unit 1)
  } else {
    az_mlp_table_decode(az_mlp_table_read(s, wave, 0, 0, lane), unused, cur);
    cur.nkb = n_tab > 0 ? cur.nkb : 0;  // 0: no unit, no loads
    // (defined on this path too: left undefined they would count as carried over from the
    // caller's previous simulation and be kept alive, or spilled, around its whole loop)
    F.w0 = F.w1 = F.w2 = az_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    F.bias0 = F.bias1 = F.bias2 = 0.0f;
  }
  return cur;
}

// (KB1 may exceed the kernel's number of slots, SLOTS: the slots beyond it do not exist)
template <int KB0, int KB1, int SLOTS = AZ_SLOTS>
AZ_DEV void az_forward_prime_slots(const MlpPre& cur, int lane, MlpRegs& R) {
  if (cur.nkb <= 0) return;
  // same order and count as a unit's refills
  const int nk = cur.nkb - 1, nk2 = cur.nkb2 > 0 ? cur.nkb2 - 1 : 0;
#pragma unroll
  for (int kb = KB0; kb < (KB1 < SLOTS ? KB1 : SLOTS); ++kb) {
    const int kc = kb < nk ? kb : nk;
    R.W0[kb] = az_ldg4(cur.w0 + (size_t)kc * 64 + lane);
    R.W1[kb] = az_ldg4(cur.w1 + (size_t)kc * 64 + lane);
    if (kb < AZ_RIDER_SLOTS) {
      const int kc2 = kb < nk2 ? kb : nk2;
      R.W2[kb < AZ_RIDER_SLOTS ? kb : 0] = az_ldg4(cur.w2 + (size_t)kc2 * 64 + lane);
    }
  }
}

template <int NWAVES, int SLOTS = AZ_SLOTS>
AZ_DEV void az_forward_prime(const ModelDev& md, const MlpLds& s, int wave, int lane, MlpRegs& R, MlpFirst& F,
                             bool allow_direct) {
  const MlpPre cur = az_forward_prime_begin<NWAVES>(s, wave, lane, F, allow_direct);
  az_forward_prime_slots<0, SLOTS, SLOTS>(cur, lane, R);
}

// The first unit in the direct form (MlpFirst): one k-block per tile out of F.
AZ_DEV void az_mlp_first_unit(const MlpUnit& u, const MlpFirst& F, float* lds_base, int lane) {
  const int lane_row = lane & 15, lane_q = lane >> 4;
  const float* ap = lds_base + u.ap_off + lane_row * u.in_stride + lane_q * u.kpq;
  const float* ap2 = lds_base + u.ap2_off + lane_row * u.in_stride2 + lane_q * u.kpq2;
  const az_f32x4 a4 = *reinterpret_cast<const az_f32x4*>(ap);
  const az_f32x4 a2 = *reinterpret_cast<const az_f32x4*>(ap2);
  az_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f}, acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, F.w0.x, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, F.w1.x, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, F.w2.x, acc2, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, F.w0.y, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, F.w1.y, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, F.w2.y, acc2, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, F.w0.z, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, F.w1.z, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, F.w2.z, acc2, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, F.w0.w, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, F.w1.w, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, F.w2.w, acc2, 0, 0, 0);
  // (a unit without a rider has computed a tile nobody stores: same instructions for every unit)
  az_mlp_unit_epilogue(u, acc0, acc1, acc2, F.bias0, F.bias1, F.bias2, lds_base, lane);
}

// Offsets of a network's head in MlpLds::head_w: [nout][kp] weights in the permuted order of the
// activations (az_act_index), then 16 biases; the policy's head first.
AZ_DEV int az_mlp_head_offset(const ModelDev& md, bool is_val) {
  const int P = az_mlp_phases(md);
  if (!is_val || md.pol.n_layers < P) return 0;
  return md.pol.nout[P - 1] * md.pol.kp[P - 1] + 16;
}

// Kernel prologue (with az_stage_inputs): the heads' weights from the packed buffer (tile 0 of
// the last layer: word ((kb * 64 + (k & 3) * 16 + col) * 4 + ((k >> 2) & 3)) holds w[k][col]).
AZ_DEV void az_mlp_stage_heads(const ModelDev& md, const MlpLds& s, int tid, int nthreads) {
  if (!az_mlp_heads_apply(md)) return;
  const int P = az_mlp_phases(md);
  for (int net = 0; net < 2; ++net) {
    const MlpLayout& L = net ? md.val : md.pol;
    if (L.n_layers < P) continue;
    const int kp = L.kp[P - 1], nout = L.nout[P - 1];
    float* dst = s.head_w + az_mlp_head_offset(md, net != 0);
    const float* w = md.packed + L.w_off[P - 1];
    for (int i = tid; i < nout * kp; i += nthreads) {
      const int col = i / kp, k = i - col * kp;
      dst[col * kp + az_act_index(k, kp)] = w[(((k >> 4) * 64 + (k & 3) * 16 + col) * 4) + ((k >> 2) & 3)];
    }
    for (int i = tid; i < 16; i += nthreads) dst[nout * kp + i] = i < nout ? md.packed[L.b_off[P - 1] + i] : 0.0f;
  }
}

// One column of an output head on the vector ALU, in the order of az_head_dot (az_math.h): the
// wave's lanes own the 16 rows (lane & 15), and each of the four groups of 16 lanes runs ONE of
// the four interleaved chains -- chain g = lane >> 4 over the inputs k = g, g + 4, ... in order --
// before two lane exchanges add them up as (c0 + c1) + (c2 + c3) (floating-point addition is
// commutative, so every group ends with the same bits).  Activations and staged weights are in
// the same permuted order, plane k & 3 holding its elements k >> 2 side by side: a chain reads
// its plane front to back, 16 bytes at a time.
template <int KP>
AZ_DEV float az_mlp_head_chain(const float* arow, const float* wcol, int kp_runtime, int lane) {
  const int kp = KP > 0 ? KP : kp_runtime, kpq = kp >> 2;
  const float* a = arow + (lane >> 4) * kpq;
  const float* w = wcol + (lane >> 4) * kpq;
  float acc = 0.0f;
  az_f32x4 a4 = *reinterpret_cast<const az_f32x4*>(a), w4 = *reinterpret_cast<const az_f32x4*>(w);
  auto body = [&](int i) {
    // (the next four elements are requested before these four are multiplied; the last turn
    // reads its own block again.  Requesting more ahead -- all operands, or groups of four
    // blocks -- was measured: the phase gets no shorter and the fused kernel starts to spill)
    const int nx = i + 4 < kpq ? i + 4 : i;
    const az_f32x4 an = *reinterpret_cast<const az_f32x4*>(a + nx), wn = *reinterpret_cast<const az_f32x4*>(w + nx);
    acc = __builtin_fmaf(a4.x, w4.x, acc);
    acc = __builtin_fmaf(a4.y, w4.y, acc);
    acc = __builtin_fmaf(a4.z, w4.z, acc);
    acc = __builtin_fmaf(a4.w, w4.w, acc);
    a4 = an;
    w4 = wn;
  };
  if constexpr (KP > 0) {
#pragma unroll
    for (int i = 0; i < KP / 4; i += 4) body(i);
  } else {
#pragma unroll 1
    for (int i = 0; i < kpq; i += 4) body(i);
  }
  const float pair = acc + __shfl_xor(acc, 16);
  return pair + __shfl_xor(pair, 32);
}

// The output heads of the last phase on the vector ALU (az_mlp_heads_apply): every output column
// of the two heads is one item, taken by one wave (az_mlp_head_chain).
AZ_DEV void az_mlp_heads(const ModelDev& md, const MlpLds& s, int wave, int lane) {
  const int phase = az_mlp_phases(md) - 1;
  const bool has_v = md.val.n_layers == phase + 1, has_p = md.pol.n_layers == phase + 1;
  const int nout_v = has_v ? md.val.nout[phase] : 0, nout_p = has_p ? md.pol.nout[phase] : 0;
  const int kp_v = has_v ? md.val.kp[phase] : 16, kp_p = has_p ? md.pol.kp[phase] : 16;
  const int in_v = (int)(((phase & 1) ? s.va : s.vb) - s.base), in_p = (int)(((phase & 1) ? s.pa : s.pb) - s.base);
  const int out_v = (int)(s.value - s.base), out_p = (int)(s.logits - s.base);
  const int n_items = nout_p + nout_v;
  const int row = lane & 15;
  for (int item = __builtin_amdgcn_readfirstlane(wave); item < n_items; item += AZ_MLP_WAVES) {
    const bool is_val = item >= nout_p;
    const int col = is_val ? item - nout_p : item;
    const int kp = is_val ? kp_v : kp_p;
    const float* arow = s.base + (is_val ? in_v : in_p) + row * (is_val ? s.vstride : s.pstride);
    const float* head = s.head_w + (is_val && has_p ? nout_p * kp_p + 16 : 0);
    const float* wcol = head + col * kp;
    const float acc = kp == 256  ? az_mlp_head_chain<256>(arow, wcol, kp, lane)
                      : kp == 128 ? az_mlp_head_chain<128>(arow, wcol, kp, lane)
                      : kp == 64  ? az_mlp_head_chain<64>(arow, wcol, kp, lane)
                                  : az_mlp_head_chain<0>(arow, wcol, kp, lane);
    const float bias = head[(is_val ? nout_v : nout_p) * kp + col];
    if (lane < 16) s.base[(is_val ? out_v + row * AZ_MAX_ATOMS : out_p + row * AZ_MAX_ACTIONS) + col] = acc + bias;
  }
}

// Both networks on the 16 rows staged in s.in0 -> s.logits / s.value.  Called by all
// NWAVES waves with the first unit's weights in R / F (az_forward_prime); every wave passes one
// workgroup barrier per phase, the last of which makes the results visible to every thread.
// With WRAP the last unit prefetches the first unit's weights again, so the next call needs
// no priming (and the direct form of the first unit is not used).
template <int NWAVES, bool WRAP, bool BF16MODE = false>
AZ_DEV void az_forward_tile(const ModelDev& md, const MlpLds& s, int wave, int lane, MlpRegs& R,
                            const MlpFirst& F) {
  const int n_tab = __builtin_amdgcn_readfirstlane(s.tab_n[wave]);
  const int lead = __builtin_amdgcn_readfirstlane(s.tab_n[NWAVES + wave]);
  az_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int stamp_i = 0;
  const int wave_ = wave;
  (void)wave_;
  AZ_STAMP_AT();
  for (int b = 0; b < lead; ++b) {
    az_lds_barrier();
  }
  // the unit whose weights the refills of unit ui fetch: the next one, or (past the last) the
  // first one again / this one once more, so that the refill loads always have a valid source
  auto ipre_of = [&](int ui) { return ui + 1 < n_tab ? ui + 1 : (WRAP ? 0 : ui); };
  const int first = !WRAP && F.direct ? 1 : 0;
  if (first) {
    MlpUnit u0;
    MlpPre unused;
    az_mlp_table_decode(az_mlp_table_read(s, wave, 0, 0, lane), u0, unused);
    AZ_STAMP_AT();
    az_mlp_first_unit(u0, F, s.base, lane);
    AZ_STAMP_AT();
    for (int b = 0; b < u0.bars; ++b) {
      az_lds_barrier();
    }
    AZ_STAMP_AT();
  }
  // The table word of a unit is read while the unit before it runs (the table does not change):
  // behind the phase barrier only its broadcast into scalar registers is left.
  uint32_t mine = n_tab > first ? az_mlp_table_read(s, wave, first, ipre_of(first), lane) : 0u;
  for (int ui = first; ui < n_tab; ++ui) {
    MlpUnit cur;
    MlpPre pre;
    az_mlp_table_decode(mine, cur, pre);
    if (!WRAP && ui == n_tab - 1) {
      // nothing follows the last unit: its refills (their number is fixed) all fetch one and the
      // same k-block instead of a whole set of weights that nobody would use
      pre.nkb = 1;
      pre.nkb2 = pre.nkb2 > 0 ? 1 : 0;
    }
    const int un = ui + 1 < n_tab ? ui + 1 : ui;
    mine = az_mlp_table_read(s, wave, un, ipre_of(un), lane);
    AZ_STAMP_AT();
    az_mlp_run_unit<BF16MODE>(cur, pre, R, acc0, acc1, s.base, lane, wave, stamp_i);
    AZ_STAMP_AT();
    for (int b = 0; b < cur.bars; ++b) {
      az_lds_barrier();
    }
    AZ_STAMP_AT();
  }
  if (az_mlp_heads_apply(md)) {
    az_mlp_heads(md, s, wave, lane);
    AZ_STAMP_AT();
    az_lds_barrier();
  }
}

#endif  // AZ_MLP_H_
