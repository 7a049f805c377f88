/* az_math.h -- the numerics specification of the search path.
 *
 * Every elementary function the search needs (exp, log, sin, cos, logistic,
 * softmax pieces, the JAX-style threefry generator and its uniform / gumbel
 * transforms) is defined here as a fixed sequence of IEEE-754 binary32
 * operations: +, -, *, /, sqrt and fused multiply-add, all correctly rounded
 * on x86-64 and on gfx950.  Compiled with -ffp-contract=off on both sides the
 * same inputs therefore give the same bits on the host and on the GPU, which
 * is what lets the discrete outputs of the search (argmax results, visit
 * counts, node indices) be compared bit for bit between a CPU run and a GPU
 * run.  The polynomial cores are the classic Cephes single-precision kernels
 * (expf, logf, sinf, cosf: < 1 ulp typical error), which is far inside the
 * 1e-5 relative tolerance allowed against a libm / XLA evaluation.
 *
 * The header is written in the common subset of C99 and HIP C++ so that a
 * plain C checker (gcc) and the device code (hipcc) include the very same
 * text.  It contains no search logic.
 */
#ifndef AZ_MATH_H_
#define AZ_MATH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ static __forceinline__
#else
#define AZ_HD static inline
#endif
/* AZ_COMPUTED_HERE(x): on the GPU, x is evaluated at this point whatever selects
 * follow (the optimiser otherwise turns "special ? constant : x" back into a
 * branch around the evaluation of x); no effect on the value. */
#if defined(__HIP_DEVICE_COMPILE__)
#define AZ_COMPUTED_HERE(x) asm volatile("" : "+v"(x))
#else
#define AZ_COMPUTED_HERE(x) ((void)0)
#endif

#define AZ_INF (__builtin_inff())
#define AZ_NAN (__builtin_nanf(""))
#define AZ_FLT_TINY 1.17549435e-38f /* smallest normal binary32 (finfo.tiny) */
#define AZ_FLT_MAX 3.40282347e+38f  /* finfo.max; finfo.min is its negation  */

AZ_HD uint32_t az_f2u(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
AZ_HD float az_u2f(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
AZ_HD float az_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
AZ_HD float az_sqrtf(float x) { return __builtin_sqrtf(x); }

// Summation order of the dense layers (part of the numerics specification: a float32 dot
// product depends on it, and the reference's `jnp.dot` fixes none).  Every output is a chain of
// fused multiply-adds over the inputs in index order, then the bias -- except the SMALL OUTPUT
// HEADS: when the deeper network's last layer (or both networks' last layers, if they are equally
// deep) has at most 16 outputs, at most 8 outputs in all and at most 1024 words of weights and
// biases (inputs padded to a multiple of 16, plus 16 bias words per head), each of its outputs
// is the sum of FOUR interleaved chains, chain g over the inputs k = g, g + 4, g + 8, ... in
// order, combined as (c0 + c1) + (c2 + c3), then the bias.  (A 256-long chain for a single value
// output is the critical path of every network evaluation otherwise; four chains of 64 are not.)
// Returns bit 0: the policy network's last layer is such a head, bit 1: the value network's.
AZ_HD int az_small_heads(int pol_layers, int pol_in, int pol_out, int val_layers, int val_in, int val_out) {
  const int P = pol_layers > val_layers ? pol_layers : val_layers;
  if (P < 2) return 0;
  int total = 0, items = 0, mask = 0;
  if (pol_layers == P) {
    if (pol_out > 16) return 0;
    total += pol_out * ((pol_in + 15) / 16 * 16) + 16;
    items += pol_out;
    mask |= 1;
  }
  if (val_layers == P) {
    if (val_out > 16) return 0;
    total += val_out * ((val_in + 15) / 16 * 16) + 16;
    items += val_out;
    mask |= 2;
  }
  return total <= 1024 && items <= 8 ? mask : 0;
}
// One output of a small head: x[k], w[k * w_stride] for k < n.
AZ_HD float az_head_dot(const float* x, const float* w, int w_stride, int n) {
  float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < n; ++k) c[k & 3] = __builtin_fmaf(x[k], w[(size_t)k * w_stride], c[k & 3]);
  return (c[0] + c[1]) + (c[2] + c[3]);
}
AZ_HD int az_isnan(float x) { return x != x; }
/* jnp.maximum / jnp.minimum propagate NaN. */
AZ_HD float az_maxf(float a, float b) {
  if (a != a) return a;
  if (b != b) return b;
  return a > b ? a : b;
}
AZ_HD float az_minf(float a, float b) {
  if (a != a) return a;
  if (b != b) return b;
  return a < b ? a : b;
}

/* 2^k for -126 <= k <= 127 built from the exponent field. */
AZ_HD float az_pow2i(int k) { return az_u2f((uint32_t)(k + 127) << 23); }

/* exp(x).  k = round(x*log2(e)) by the add-magic-constant trick, Cody-Waite
 * reduction r = x - k*ln2 in two fused steps, degree-5 polynomial on r^2 plus
 * the linear terms, and a two-step power-of-two scaling so that results in the
 * subnormal range are rounded once. */
AZ_HD float az_expf(float x) {
  /* Branch-free on purpose (selects, no early returns): on the GPU this runs in
   * the epilogue of every dense layer, where divergent branches cost more than
   * the arithmetic.  Out-of-range and NaN inputs are computed on a stand-in and
   * replaced at the end. */
  const int is_nan = x != x;
  const int over = x > 88.72283935546875f;
  const int under = x < -103.97208404541015625f;
  const float xs = (is_nan | over | under) ? 0.0f : x;
  float kf = (xs * 1.44269502162933349609375f + 12582912.0f) - 12582912.0f;
  float r = az_fmaf(kf, -0.693359375f, xs);
  r = az_fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = az_fmaf(p, r, 1.3981999507e-3f);
  p = az_fmaf(p, r, 8.3334519073e-3f);
  p = az_fmaf(p, r, 4.1665795894e-2f);
  p = az_fmaf(p, r, 1.6666665459e-1f);
  p = az_fmaf(p, r, 5.0000001201e-1f);
  p = az_fmaf(p, r * r, r);
  p = p + 1.0f;
  int k = (int)kf;
  int k1 = k >> 1;
  int k2 = k - k1;
  float y = (p * az_pow2i(k1)) * az_pow2i(k2);
  y = under ? 0.0f : y;
  y = over ? AZ_INF : y;
  return is_nan ? x : y;
}

/* log(x): frexp by bit manipulation (subnormals pre-scaled by 2^23), the
 * Cephes degree-8 polynomial on m - 1 with m in [sqrt(1/2), sqrt(2)). */
AZ_HD float az_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return AZ_NAN;
  if (x == 0.0f) return -AZ_INF;
  if (x == AZ_INF) return x;
  int e = 0;
  uint32_t ux = az_f2u(x);
  if (ux < 0x00800000u) {
    x = x * 8388608.0f;
    e = -23;
    ux = az_f2u(x);
  }
  e += (int)(ux >> 23) - 126;
  float m = az_u2f((ux & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106769084930419921875f) {
    e -= 1;
    m = (m + m) - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float y = 7.0376836292e-2f;
  y = az_fmaf(y, m, -1.1514610310e-1f);
  y = az_fmaf(y, m, 1.1676998740e-1f);
  y = az_fmaf(y, m, -1.2420140846e-1f);
  y = az_fmaf(y, m, 1.4249322787e-1f);
  y = az_fmaf(y, m, -1.6668057665e-1f);
  y = az_fmaf(y, m, 2.0000714765e-1f);
  y = az_fmaf(y, m, -2.4999993993e-1f);
  y = az_fmaf(y, m, 3.3333331174e-1f);
  y = (y * m) * z;
  float fe = (float)e;
  y = az_fmaf(fe, -2.12194440e-4f, y);
  y = az_fmaf(-0.5f, z, y);
  float res = m + y;
  res = az_fmaf(fe, 0.693359375f, res);
  return res;
}

/* Shared octant reduction for sin / cos: returns the reduced argument in
 * [-pi/4, pi/4] and the octant index j (even, 0..6).  |x| >= 2^30 has no
 * significant fraction left in binary32 and is mapped to 0 * x. */
AZ_HD float az_trig_reduce_(float ax, int* j_out) {
  int j = (int)(ax * 1.27323949337005615234375f); /* 4/pi */
  float y = (float)j;
  if (j & 1) {
    j += 1;
    y += 1.0f;
  }
  *j_out = j & 7;
  float r;
  if (ax > 8192.0f) {
    r = az_fmaf(y, -0.785398185253143310546875f, ax);
  } else {
    r = az_fmaf(y, -0.78515625f, ax);
    r = az_fmaf(y, -2.4187564849853515625e-4f, r);
    r = az_fmaf(y, -3.77489497744594108e-8f, r);
  }
  return r;
}
AZ_HD float az_sin_poly_(float r) {
  float z = r * r;
  float p = -1.9515295891e-4f;
  p = az_fmaf(p, z, 8.3321608736e-3f);
  p = az_fmaf(p, z, -1.6666654611e-1f);
  return az_fmaf(p * z, r, r);
}
AZ_HD float az_cos_poly_(float r) {
  float z = r * r;
  float p = 2.443315711809948e-5f;
  p = az_fmaf(p, z, -1.388731625493765e-3f);
  p = az_fmaf(p, z, 4.166664568298827e-2f);
  float y = (p * z) * z;
  y = az_fmaf(-0.5f, z, y);
  return y + 1.0f;
}
AZ_HD float az_sinf(float x) {
  if (x != x) return x;
  float ax = x < 0.0f ? -x : x;
  if (!(ax < 1073741824.0f)) return 0.0f * x;
  int j;
  float r = az_trig_reduce_(ax, &j);
  int neg = x < 0.0f;
  if (j > 3) {
    neg = !neg;
    j -= 4;
  }
  float y = (j == 2) ? az_cos_poly_(r) : az_sin_poly_(r);
  return neg ? -y : y;
}
AZ_HD float az_cosf(float x) {
  if (x != x) return x;
  float ax = x < 0.0f ? -x : x;
  if (!(ax < 1073741824.0f)) return 0.0f * x + (ax == AZ_INF ? 0.0f : 1.0f);
  int j;
  float r = az_trig_reduce_(ax, &j);
  int neg = 0;
  if (j > 3) {
    neg = 1;
    j -= 4;
  }
  if (j > 1) neg = !neg;
  float y = (j == 2) ? az_sin_poly_(r) : az_cos_poly_(r);
  return neg ? -y : y;
}

/* The same two functions in ONE body without branches or early returns (selects
 * only; special inputs run on the stand-in 0 and are replaced at the end, as in
 * az_expf): az_trig_(x, 0) == az_sinf(x) and az_trig_(x, 1) == az_cosf(x) bit for
 * bit, NaN payloads included (compared on every 7th binary32 pattern).  For
 * callers that evaluate several sines / cosines side by side -- straight-line
 * code lets their instructions interleave, and a group of GPU threads can
 * evaluate a mix of both in one pass.  Not a replacement for the forms above: in
 * register-bound kernels their branches keep fewer values live. */
AZ_HD float az_trig_reduce_sel_(float ax, int* j_out) {
  int j = (int)(ax * 1.27323949337005615234375f); /* 4/pi */
  const float y0 = (float)j;
  const int odd = j & 1;
  const float y = odd ? y0 + 1.0f : y0;
  j += odd;
  *j_out = j & 7;
  const float r_far = az_fmaf(y, -0.785398185253143310546875f, ax);
  float r = az_fmaf(y, -0.78515625f, ax);
  r = az_fmaf(y, -2.4187564849853515625e-4f, r);
  r = az_fmaf(y, -3.77489497744594108e-8f, r);
  return ax > 8192.0f ? r_far : r;
}
/* sin (want_cos == 0) or cos of x; one body for both so that a group of GPU
 * threads can evaluate a mix of sines and cosines in one pass (with a constant
 * flag it folds to the plain function).  |x| >= 2^30 has no significant
 * fraction left in binary32: sin gives 0 * x (NaN for an infinity), cos 1 (NaN
 * for an infinity).  NaN returns itself. */
AZ_HD float az_trig_(float x, int want_cos) {
  const int is_nan = x != x;
  const float ax0 = x < 0.0f ? -x : x;
  const int far = !(ax0 < 1073741824.0f); /* also true for NaN */
  const float ax = far ? 0.0f : ax0;
  int j;
  const float r = az_trig_reduce_sel_(ax, &j);
  int neg = want_cos ? 0 : (x < 0.0f);
  const int upper = j > 3;
  neg = upper ? !neg : neg;
  j = upper ? j - 4 : j;
  neg = (want_cos && j > 1) ? !neg : neg;
  const float ys = az_sin_poly_(r), yc = az_cos_poly_(r);
  const int cos_poly = want_cos ? (j != 2) : (j == 2);
  float y = cos_poly ? yc : ys;
  y = neg ? -y : y;
  AZ_COMPUTED_HERE(y);
  const float far_sin = 0.0f * x;
  const float far_cos = 0.0f * x + (ax0 == AZ_INF ? 0.0f : 1.0f);
  y = far ? (want_cos ? far_cos : far_sin) : y;
  return is_nan ? x : y;
}

/* logistic(x) = 1 / (1 + exp(t)) with t = -x clamped to [-64, 64];  swish(x) = x * logistic(x).
 * The clamp is part of the specification: it moves the result by less than 2e-28 (logistic is
 * below 1.6e-28 for x < -64 and rounds to 1 for x > 17), and it keeps exp(t), 1 + exp(t) and
 * the quotient normal numbers far from over- and underflow, so that the GPU's packed form
 * (az_swish.h) needs no range handling at all.  A NaN passes through the clamp. */
AZ_HD float az_logistic(float x) {
  float t = -x;
  t = t < -64.0f ? -64.0f : t;
  t = t > 64.0f ? 64.0f : t;
  return 1.0f / (1.0f + az_expf(t));
}
AZ_HD float az_swish(float x) { return x * az_logistic(x); }
/* x ** e for x >= 0 as exp(e * log(x)); e == 0 gives exactly 1 (also for
 * x == 0), matching the power function's convention. */
AZ_HD float az_powf(float x, float e) {
  if (e == 0.0f) return 1.0f;
  if (e == 1.0f) return x;
  return az_expf(e * az_logf(x));
}

/* ------------------------------------------------------------------ */
/* Threefry-2x32, 20 rounds (Salmon et al. 2011), the block function of
 * JAX's default PRNG.  key = (k0, k1), counter = (x0, x1). */
AZ_HD uint32_t az_rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

AZ_HD void az_threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                           uint32_t* o0, uint32_t* o1) {
  uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += ks0;
  x1 += ks1;
#define AZ_TF_ROUND(r) \
  x0 += x1;            \
  x1 = az_rotl32(x1, r); \
  x1 ^= x0;
  AZ_TF_ROUND(13) AZ_TF_ROUND(15) AZ_TF_ROUND(26) AZ_TF_ROUND(6)
  x0 += ks1; x1 += ks2 + 1u;
  AZ_TF_ROUND(17) AZ_TF_ROUND(29) AZ_TF_ROUND(16) AZ_TF_ROUND(24)
  x0 += ks2; x1 += ks0 + 2u;
  AZ_TF_ROUND(13) AZ_TF_ROUND(15) AZ_TF_ROUND(26) AZ_TF_ROUND(6)
  x0 += ks0; x1 += ks1 + 3u;
  AZ_TF_ROUND(17) AZ_TF_ROUND(29) AZ_TF_ROUND(16) AZ_TF_ROUND(24)
  x0 += ks1; x1 += ks2 + 4u;
  AZ_TF_ROUND(13) AZ_TF_ROUND(15) AZ_TF_ROUND(26) AZ_TF_ROUND(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef AZ_TF_ROUND
  *o0 = x0;
  *o1 = x1;
}

/* Key derivation and bit generation in the "partitionable" counter layout
 * (the default of JAX >= 0.5): element i of a split / of a 32-bit draw uses
 * the 64-bit counter (hi = 0, lo = i); a split keeps both output words as the
 * new key, a 32-bit draw xors them. */
AZ_HD void az_key_split(uint32_t k0, uint32_t k1, uint32_t i, uint32_t* o0,
                        uint32_t* o1) {
  az_threefry2x32(k0, k1, 0u, i, o0, o1);
}
AZ_HD uint32_t az_random_bits32(uint32_t k0, uint32_t k1, uint32_t i) {
  uint32_t a, b;
  az_threefry2x32(k0, k1, 0u, i, &a, &b);
  return a ^ b;
}
/* 32 random bits -> float in [0, 1): mantissa fill of 1.0, minus 1. */
AZ_HD float az_bits_to_unit(uint32_t bits) {
  return az_u2f((bits >> 9) | 0x3f800000u) - 1.0f;
}
/* uniform(minval, maxval) from bits: f*(max-min)+min clamped below by min. */
AZ_HD float az_uniform_from_bits(uint32_t bits, float minval, float maxval) {
  float f = az_bits_to_unit(bits);
  float v = f * (maxval - minval) + minval;
  return v > minval ? v : minval;
}
/* standard Gumbel from bits: -log(-log(u)), u ~ uniform(tiny, 1). */
AZ_HD float az_gumbel_from_bits(uint32_t bits) {
  float u = az_uniform_from_bits(bits, AZ_FLT_TINY, 1.0f);
  return -az_logf(-az_logf(u));
}

#endif /* AZ_MATH_H_ */
