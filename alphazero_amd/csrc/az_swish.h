// az_swish.h -- az_swish (az_math.h) on packed pairs, for the epilogue of the dense layers.
//
// The epilogue is vector-ALU bound (two waves per SIMD evaluate 12 activations per thread
// while the matrix pipe idles), so the instruction count matters.  Every element gets the
// IEEE operations of az_swish in the same order, so the bits are the same as on the host;
// what differs is how they are issued:
//   * multiplies, adds and fused multiply-adds work on two values per instruction;
//   * the specification clamps the exponent's argument to [-64, 64], so none of az_expf's
//     range selects can fire and they are left out;
//   * 2^k is built from the integer that the round-to-nearest trick leaves in the low
//     mantissa bits of t*log2(e) + 1.5*2^23, and applied in one multiply (p * 2^k1 * 2^k2 and
//     p * 2^k are both exact, hence equal, when the result is a normal number);
//   * 1 / (1 + e) is the compiler's own correctly rounded division sequence (reciprocal
//     estimate, one Newton step, two quotient corrections) without its range-scaling steps
//     -- they are the identity for a numerator of 1 and a denominator in [1, 2^93] -- and
//     with its fused multiply-adds packed.
// There is no branch in here on purpose: the code sits between the matrix loop and the
// refills of ~150 live weight registers.
#ifndef AZ_SWISH_H_
#define AZ_SWISH_H_

#include "az_device.h"

typedef __attribute__((ext_vector_type(2))) float az_f32x2;

AZ_DEV az_f32x2 az_fma2(az_f32x2 a, az_f32x2 b, az_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
AZ_DEV az_f32x2 az_splat2(float v) { return az_f32x2{v, v}; }

// 1 + exp(t), t = clamp(-x, -64, 64), both elements.
AZ_DEV az_f32x2 az_one_plus_exp_neg2(az_f32x2 x) {
  // the median of (t, -64, 64) is the clamp; a NaN comes out as a number here, and the
  // final multiplication by x makes the result a NaN again, as the specification's does
  const az_f32x2 t = {__builtin_amdgcn_fmed3f(-x.x, -64.0f, 64.0f), __builtin_amdgcn_fmed3f(-x.y, -64.0f, 64.0f)};
  const az_f32x2 m = t * az_splat2(1.44269502162933349609375f) + az_splat2(12582912.0f);
  const az_f32x2 kf = m - az_splat2(12582912.0f);
  az_f32x2 r = az_fma2(kf, az_splat2(-0.693359375f), t);
  r = az_fma2(kf, az_splat2(2.12194440e-4f), r);
  az_f32x2 p = az_splat2(1.9875691500e-4f);
  p = az_fma2(p, r, az_splat2(1.3981999507e-3f));
  p = az_fma2(p, r, az_splat2(8.3334519073e-3f));
  p = az_fma2(p, r, az_splat2(4.1665795894e-2f));
  p = az_fma2(p, r, az_splat2(1.6666665459e-1f));
  p = az_fma2(p, r, az_splat2(5.0000001201e-1f));
  p = az_fma2(p, r * r, r);
  p = p + az_splat2(1.0f);
  // m = 1.5 * 2^23 + k exactly, so its low mantissa bits hold k (two's complement);
  // (bits << 23) + bits(1.0f) is the float 2^k for |k| <= 126
  const az_f32x2 s = {az_u2f((az_f2u(m.x) << 23) + 0x3f800000u), az_u2f((az_f2u(m.y) << 23) + 0x3f800000u)};
  return az_splat2(1.0f) + p * s;
}

// 1 / d, correctly rounded, for d in [1, 2^100], both elements.
AZ_DEV az_f32x2 az_rcp2_exact(az_f32x2 d) {
  const az_f32x2 one = az_splat2(1.0f);
  az_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  r = az_fma2(az_fma2(-d, r, one), r, r);
  az_f32x2 q = r;
  q = az_fma2(az_fma2(-d, q, one), r, q);
  return az_fma2(az_fma2(-d, q, one), r, q);
}

AZ_DEV az_f32x2 az_swish2(az_f32x2 x) { return x * az_rcp2_exact(az_one_plus_exp_neg2(x)); }

// az_swish2 on four pairs at once, step by step across the four: every instruction has three
// independent ones between itself and its consumer, so the sequence runs at the issue rate
// instead of at the latency of one dependent chain.  The empty asm after each step keeps
// the order (left alone, the scheduler -- short of registers next to the weight slots --
// runs the chains nearly one after the other).  Same operations per element as az_swish2.
// N = 2: the two pairs of a single tile (a rider's), which otherwise run one after the other.
// (Six at once -- a rider's two with the pair's four -- was measured: the fused kernel spills
// three registers and gets 2 % slower.)
template <int N>
AZ_DEV void az_sw_keep(az_f32x2 (&a)[N]) {
  if constexpr (N == 4)
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  else
    asm volatile("" : "+v"(a[0]), "+v"(a[1]));
}
#define AZ_SW4(arr, expr)                                         \
  _Pragma("unroll") for (int i = 0; i < N; ++i) arr[i] = (expr); \
  az_sw_keep<N>(arr)
template <int N>
AZ_DEV void az_swish2xn(az_f32x2 (&x)[N]) {
  static_assert(N == 4 || N == 2, "four or two pairs");
  const az_f32x2 one = az_splat2(1.0f);
  az_f32x2 t[N], m[N], kf[N], r[N], rr[N], p[N], sc[N], d[N], y[N], e[N], q[N];
  // 1 + exp(t), as in az_one_plus_exp_neg2
  AZ_SW4(t, (az_f32x2{__builtin_amdgcn_fmed3f(-x[i].x, -64.0f, 64.0f), __builtin_amdgcn_fmed3f(-x[i].y, -64.0f, 64.0f)}));
  AZ_SW4(m, t[i] * az_splat2(1.44269502162933349609375f));
  AZ_SW4(m, m[i] + az_splat2(12582912.0f));
  AZ_SW4(kf, m[i] - az_splat2(12582912.0f));
  AZ_SW4(r, az_fma2(kf[i], az_splat2(-0.693359375f), t[i]));
  AZ_SW4(r, az_fma2(kf[i], az_splat2(2.12194440e-4f), r[i]));
  AZ_SW4(p, az_fma2(az_splat2(1.9875691500e-4f), r[i], az_splat2(1.3981999507e-3f)));
  AZ_SW4(p, az_fma2(p[i], r[i], az_splat2(8.3334519073e-3f)));
  AZ_SW4(p, az_fma2(p[i], r[i], az_splat2(4.1665795894e-2f)));
  AZ_SW4(p, az_fma2(p[i], r[i], az_splat2(1.6666665459e-1f)));
  AZ_SW4(p, az_fma2(p[i], r[i], az_splat2(5.0000001201e-1f)));
  AZ_SW4(rr, r[i] * r[i]);
  AZ_SW4(p, az_fma2(p[i], rr[i], r[i]));
  AZ_SW4(p, p[i] + one);
  AZ_SW4(sc, (az_f32x2{az_u2f((az_f2u(m[i].x) << 23) + 0x3f800000u), az_u2f((az_f2u(m[i].y) << 23) + 0x3f800000u)}));
  AZ_SW4(p, p[i] * sc[i]);
  AZ_SW4(d, one + p[i]);
  // 1 / d, as in az_rcp2_exact
  AZ_SW4(y, (az_f32x2{__builtin_amdgcn_rcpf(d[i].x), __builtin_amdgcn_rcpf(d[i].y)}));
  AZ_SW4(e, az_fma2(-d[i], y[i], one));
  AZ_SW4(y, az_fma2(e[i], y[i], y[i]));
  AZ_SW4(e, az_fma2(-d[i], y[i], one));
  AZ_SW4(q, az_fma2(e[i], y[i], y[i]));
  AZ_SW4(e, az_fma2(-d[i], q[i], one));
  AZ_SW4(q, az_fma2(e[i], y[i], q[i]));
  AZ_SW4(x, x[i] * q[i]);
}
#undef AZ_SW4
AZ_DEV void az_swish2x4(az_f32x2 (&x)[4]) { az_swish2xn<4>(x); }

#endif  // AZ_SWISH_H_
