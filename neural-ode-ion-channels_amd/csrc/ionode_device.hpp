// ionode_device.hpp -- gfx950 (CDNA4, MI355X) device code of the batched dopri5 integrator.
//
// Execution model (DESIGN.md "Kernels"):
//   * A *tile* of trajectories advances in lock-step over step ATTEMPTS (every attempt costs the
//     same six RHS evaluations whether it is accepted or not), each trajectory with its own
//     t, dt and accept/reject decision.  Closed-form models: 64 trajectories per wavefront, one
//     per lane.  MLP models (NN-f / NN-d): 16 trajectories per tile, lane = 16*q + j holds
//     trajectory j (replicated over q = 0..3 and over the G wavefronts of the workgroup), which
//     is exactly the B-operand / accumulator column layout of v_mfma_f32_16x16x4_f32.
//   * The stage MLP of the tile is a chain of [NP x NP] x [NP x 16] products on the fp32 MFMA
//     (bit-for-bit an fmaf chain, i.e. the reference's fp32 arithmetic; same rate as the fp32 VALU
//     but one VGPR per operand and no broadcast traffic).  Accumulator tiles are handed to the
//     next layer as B operands without any transpose by permuting the contraction index:
//     register r of lane-group q of tile kt is k = 16*kt + 4*q + r.  Weights stream from L2 in
//     that fragment order (host-packed, 1 KiB per wave-load); activations are exchanged between
//     the G wavefronts through LDS (double-buffered, one barrier per layer).
//   * Dense output is emitted cooperatively: for every trajectory whose step was accepted, the
//     wavefront evaluates the 4th-order interpolant at 64 consecutive output times at once and
//     stores them coalesced (D*sizeof(S) bytes per lane).
//
// Arithmetic follows torchdiffeq 0.2.1's dopri5 operation by operation in the
// dtype torch would use; this translation unit is compiled with -ffp-contract=off so every a*b+c in
// the source is two IEEE operations, and the only fused operations are the explicit fmaf()/MFMA
// chains of the MLP.  Reference RHS definitions: train-s1.py:161-177 (HH), train-d1.py:165-187
// (6-state), train-s1.py:231-247 (NN-f), train-d2.py:247-272 (NN-d); protocol rule train-s1.py:218-237.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/ionode.h"

namespace ionode {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct KArgs {
  const float *mlp;  // packed image (ionode_mlp_pack)
  const double *params;
  const double *prot_v;
  const double *prot_t;
  const int32_t *prot_of_traj;
  const void *y0;
  const double *t_eval;
  void *y_out;
  double *i_out;
  int32_t *status;
  int64_t *stats;
  int32_t B, Nt, P, Np, n_params, L, N, NP, NT;
  int64_t max_steps;   // attempts allowed between two emitted outputs (torchdiffeq's max_num_steps: _advance resets its counter)
  int64_t max_total;   // attempts allowed over the whole solve (runaway bound; the reference uses a 600 s SIGALRM, train-d0.py:309-318)
  double *ckpt;        // optional [B][ckpt_cap][4 + 8*D] fp64: one record per ACCEPTED step, for the backward sweep (ionode_grad.hpp)
  int32_t ckpt_cap;
  double prot_t0, prot_dt, v_oob, rtol, atol, obs_g, obs_e;
  double prot_rdt;     // 1.0 / prot_dt, correctly rounded (host division): divisions by prot_dt become div_by()
  double dt_max;       // optional cap on the step size (+inf: none -- torchdiffeq 0.2.1 has no such option)
  int32_t obs_open;
  double *step_log;
  int64_t step_log_cap;
  const double *sse_ref;  // fused objective (hint path): reference currents [P][Nt]; per trajectory sum_k (i_k - ref[prot][k])^2 ...
  double *sse_out;        // ... goes to sse_out[B] (inf for failed trajectories); y_out / i_out may then be NULL: no trace leaves the chip
  double te_t0, te_dt;  // hint: t_eval[k] ~ te_t0 + k*te_dt (te_dt <= 0: no hint).  Only ever a guess; see emit.
  double te_rdt;        // 1 / te_dt (for the guess only)
  int32_t te_exact;     // 1: the caller VERIFIED t_eval[k] == te_t0 + (double)k * te_dt bit for bit (fp64 multiply, then add):
                        // closed-form kernels then form output times arithmetically -- no vector load sits behind their stores
  const double *v_tab;  // optional [P][Nt]: protocol voltage AT the output times (ionode_protocol_at_outputs); the closed-form
                        // kernels' current / objective epilogue then loads V(t_k) instead of re-deriving it per trajectory
  int64_t mlp_stride;   // several weight images (an ensemble / a population of nets): floats between consecutive images ...
  int32_t traj_per_img; // ... and how many consecutive trajectories share one (a multiple of the tile size); 0: one image for all
  int32_t lw_bytes;     // lane-wise kernels: LDS bytes of ONE wavefront's region (a workgroup carries four of them, see the kernel)
  const int32_t *order; // optional launch order (a permutation of 0..B-1): launch slot s integrates trajectory order[s]; every
                        // input and output stays at the trajectory's own index -- only the tiling / lane assignment changes
};

// Dormand-Prince / Shampine coefficients.
__device__ constexpr double kAlpha[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
__device__ constexpr double kBeta[6][6] = {
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
};
__device__ constexpr double kCerr[7] = {
    35.0 / 384 - 1951.0 / 21600,       0.0,
    500.0 / 1113 - 22642.0 / 50085,    125.0 / 192 - 451.0 / 720,
    -2187.0 / 6784 - -12231.0 / 42400, 11.0 / 84 - 649.0 / 6300,
    -1.0 / 60.0,
};
__device__ constexpr double kCmid[7] = {
    6025192743.0 / 30085553152.0 / 2,     0.0,
    51252292925.0 / 65400821598.0 / 2,    -2691868925.0 / 45128329728.0 / 2,
    187940372067.0 / 1594534317056.0 / 2, -1776094331.0 / 19743644256.0 / 2,
    11237099.0 / 235043384.0 / 2,
};

// Deterministic exp() and fifth root (DESIGN.md "Deterministic transcendentals"): dopri5's controller
// amplifies last-ulp differences of these two functions chaotically, so results are only reproducible
// across devices/libraries if both are fixed IEEE operation sequences.  < 1 ulp / <= 2 ulp accurate.
__device__ __forceinline__ double pow2i(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }

// fma(p, r, c) with the constant c as a SCALAR operand.  Left to itself hipcc emits v_fmac_f64 with c copied into the destination
// register first (two v_mov_b32 per constant and use -- or, with machine-LICM, every constant hoisted into a VGPR pair that is
// then spilled); both are VALU instructions on the pipe the f32 MFMA shares.  An SGPR pair costs two s_mov_b32.
__device__ __forceinline__ double fma_sc(double p, double r, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
  return d;
}

__device__ __forceinline__ double det_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return __builtin_inf();
  if (x < -745.1332191019412) return 0.0;
  const double kf = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kf, 0x1.62e42fee00000p-1, x);
  r = fma(-kf, 0x1.a39ef35793c76p-33, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int k = (int)kf;
  const int k1 = k / 2;
  return (p * pow2i(k1)) * pow2i(k - k1);
}
__device__ __forceinline__ float det_expf(float x) { return (float)det_exp((double)x); }

// The same function for the closed-form kernels' stage loop, where exp is a third of the issue work: the two power-of-two
// multiplications (p * 2^k1) * 2^(k - k1) are one v_ldexp_f64.  Bit-identical: p * 2^k is exact while the result is normal,
// and where it is subnormal or overflows both forms round exactly once (the first factor of the product form is always exact).
__device__ __forceinline__ double det_exp_ldexp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return __builtin_inf();
  if (x < -745.1332191019412) return 0.0;
  const double kf = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kf, 0x1.62e42fee00000p-1, x);
  r = fma(-kf, 0x1.a39ef35793c76p-33, r);
  // addend constants as SCALAR operands (fma_sc): left to itself hipcc writes each of them into a VGPR pair first (v_fmac_f64 has
  // its addend tied to the destination) -- 20 v_mov_b32 per call, a third of the closed-form stage loop's vector instructions
  double p = 1.0 / 6227020800.0;
  p = fma_sc(p, r, 1.0 / 479001600.0);
  p = fma_sc(p, r, 1.0 / 39916800.0);
  p = fma_sc(p, r, 1.0 / 3628800.0);
  p = fma_sc(p, r, 1.0 / 362880.0);
  p = fma_sc(p, r, 1.0 / 40320.0);
  p = fma_sc(p, r, 1.0 / 5040.0);
  p = fma_sc(p, r, 1.0 / 720.0);
  p = fma_sc(p, r, 1.0 / 120.0);
  p = fma_sc(p, r, 1.0 / 24.0);
  p = fma_sc(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)kf);
}

// det_exp_ldexp() for arguments already known to be in [-708, 708] (no NaN, no overflow, result normal): the same operation
// sequence without the three range cases.  closed_rates() tests the wavefront's arguments of a stage together (one compare each,
// one ballot) and takes this path when every lane qualifies -- always, on physical parameters.
__device__ __forceinline__ double det_exp_inrange(double x) {
  const double kf = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kf, 0x1.62e42fee00000p-1, x);
  r = fma(-kf, 0x1.a39ef35793c76p-33, r);
  double p = 1.0 / 6227020800.0;
  p = fma_sc(p, r, 1.0 / 479001600.0);
  p = fma_sc(p, r, 1.0 / 39916800.0);
  p = fma_sc(p, r, 1.0 / 3628800.0);
  p = fma_sc(p, r, 1.0 / 362880.0);
  p = fma_sc(p, r, 1.0 / 40320.0);
  p = fma_sc(p, r, 1.0 / 5040.0);
  p = fma_sc(p, r, 1.0 / 720.0);
  p = fma_sc(p, r, 1.0 / 120.0);
  p = fma_sc(p, r, 1.0 / 24.0);
  p = fma_sc(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)kf);
}

// det_exp for the MLP kernels' rate terms: same operation sequence, constants as scalar operands (fma_sc), scaling by one
// v_ldexp_f64 (bit-identical, det_exp_ldexp below) -- 36 instead of 62 vector instructions per call.
__device__ __forceinline__ double det_exp_s(double x) {
  // branch-free: the range cases are selects behind the polynomial (in-range arguments take the same operations as det_exp;
  // out-of-range arguments compute a discarded value), so two calls interleave instead of running under exec masks
  const double kf = rint(x * 0x1.71547652b82fep+0);
  double r = fma(-kf, 0x1.62e42fee00000p-1, x);
  r = fma(-kf, 0x1.a39ef35793c76p-33, r);
  double p = 1.0 / 6227020800.0;
  p = fma_sc(p, r, 1.0 / 479001600.0);
  p = fma_sc(p, r, 1.0 / 39916800.0);
  p = fma_sc(p, r, 1.0 / 3628800.0);
  p = fma_sc(p, r, 1.0 / 362880.0);
  p = fma_sc(p, r, 1.0 / 40320.0);
  p = fma_sc(p, r, 1.0 / 5040.0);
  p = fma_sc(p, r, 1.0 / 720.0);
  p = fma_sc(p, r, 1.0 / 120.0);
  p = fma_sc(p, r, 1.0 / 24.0);
  p = fma_sc(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  double e = __builtin_ldexp(p, (int)kf);  // == (p * 2^k1) * 2^(k - k1), see det_exp_ldexp
  e = (x > 709.782712893384) ? __builtin_inf() : e;
  e = (x < -745.1332191019412) ? 0.0 : e;
  return (x != x) ? x : e;
}

// a / b when rb = RN(1 / b) is at hand: q0 = RN(a * rb) is a faithful quotient, and one correction step with the exact
// remainder r = a - b*q0 (fma) gives RN(a / b) -- the correctly rounded IEEE quotient, bit for bit what `a / b` returns
// (Markstein 1990; holds barring over/underflow: operands here are times in ms, voltages in mV and O(1) ratios).  Three
// fp64 VALU operations instead of the ~12-instruction v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup sequence;
// the divisions by prot_dt (2 per protocol lookup), by 5 (det_root5) and by the step length (every dense-output sample)
// were ~3/4 of the closed-form kernels' issue work.  Zero, infinite and NaN quotients are passed through unchanged
// (the correction would turn inf into NaN and lose the sign of a zero).
__device__ __forceinline__ double div_by(double a, double b, double rb) {
  const double q0 = a * rb;
  const double r = fma(-b, q0, a);
  const double q1 = fma(r, rb, q0);
  const double aq = __builtin_fabs(q0);
  return (aq > 0.0 && aq < __builtin_inf()) ? q1 : q0;
}

// a / b for 0 <= a <= b with b a finite positive step length (every dense-output sample: x = (t_k - t0) / (t1 - t0), t_k in
// (t0, t1]): the quotient is 0 or in [2^-70, 1], so the pass-through guard of div_by() -- four vector instructions of the ~45 a
// dense-output sample costs -- is dead weight.  A zero stays +0 through both fma.
#ifndef IONODE_DIV_POS
#define IONODE_DIV_POS 1
#endif
__device__ __forceinline__ double div_pos(double a, double b, double rb) {
#if IONODE_DIV_POS
  const double q0 = a * rb;
  return fma(fma(-b, q0, a), rb, q0);
#else
  return div_by(a, b, rb);
#endif
}

// a / b for a compile-time constant b (rb = RN(1 / b)): the same correction step, with the true division kept for the quotients
// the proof excludes (zero, subnormal range, overflow).  fp32: checked against x / 1000.0f for all 2^32 inputs -- they differ only
// where |quotient| < 2^-126 (67 108 inputs, all |x| < 9.5e-38); the guards below are far inside the safe range.
__device__ __forceinline__ double div_const(double a, double b, double rb) {
  const double q0 = a * rb;
  const double r = fma(-b, q0, a);
  const double q1 = fma(r, rb, q0);
  const double aq = __builtin_fabs(q0);
  return (aq > 0x1p-900 && aq < 0x1p+900) ? q1 : a / b;
}
__device__ __forceinline__ float div_constf(float a, float b, float rb) {
  const float q0 = a * rb;
  const float r = fmaf(-b, q0, a);
  const float q1 = fmaf(r, rb, q0);
  const float aq = __builtin_fabsf(q0);
  return (aq > 0x1p-100f && aq < 0x1p+100f) ? q1 : a / b;
}

__device__ __forceinline__ double det_root5(double x) {
  if (!(x < __builtin_inf()) || !(x > 0.0)) return x;
  unsigned long long u = (unsigned long long)__double_as_longlong(x);
  u = u / 5ull + 0x3325999999999999ull;
  double y = __longlong_as_double((long long)u);
#pragma unroll
  for (int it = 0; it < 7; ++it) {
    const double y2 = y * y;
    const double y4 = y2 * y2;
    y = div_by(4.0 * y + x / y4, 5.0, 0.2);  // 0.2 == RN(1/5)
  }
  return y;
}

template <typename S> struct Real;
template <> struct Real<float> {
  static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float prev_(float x) { return nextafterf(x, x - 1.0f); }
};
template <> struct Real<double> {
  static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
  static __device__ __forceinline__ double prev_(double x) { return nextafter(x, x - 1.0); }
};

// Diagnostic build only (make EXTRA=-DIONODE_STAMPS): s_memtime phase stamps of workgroup 0 / wavefront 0, summed
// in SGPR-side 64-bit counters and written to step_log[0..15] at kernel end (no stamp executes in the real build).
#ifdef IONODE_STAMPS
struct Stamps {
  unsigned long long acc[16];
  unsigned long long last;
};
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define STAMP_DECL Stamps stamps_; for (int i_ = 0; i_ < 16; ++i_) stamps_.acc[i_] = 0; stamps_.last = stamp_now();
#define STAMP(st, slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = stamp_now(); (st).acc[slot] += n_ - (st).last; (st).last = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP_DECL
#define STAMP(st, slot) do { } while (0)
#endif

template <int MODEL> struct ModelTraits {
  static constexpr int D = (MODEL == IONODE_MODEL_MARKOV6) ? 6 : 2;
  static constexpr int NPAR = (MODEL == IONODE_MODEL_MARKOV6) ? 12 : 8;
  static constexpr bool MLP = (MODEL == IONODE_MODEL_NNF || MODEL == IONODE_MODEL_NND);
};

__device__ __forceinline__ double bcast_f64(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast_f32(float x, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
template <typename S> __device__ __forceinline__ S bcast(S x, int src);
template <> __device__ __forceinline__ double bcast<double>(double x, int src) { return bcast_f64(x, src); }
template <> __device__ __forceinline__ float bcast<float>(float x, int src) { return bcast_f32(x, src); }

// x moved across lanes by a DPP row operation (VALU speed; __shfl_xor takes two LDS-crossbar round trips for a double)
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Sum over each group of 8 consecutive lanes (3 DPP steps); every lane of the group holds the group's sum.
__device__ __forceinline__ double group8_sum_f64(double x) {
  x = x + dpp_f64<0xB1, 0xf>(x);   // quad_perm [1,0,3,2]
  x = x + dpp_f64<0x4E, 0xf>(x);   // quad_perm [2,3,0,1]
  x = x + dpp_f64<0x141, 0xf>(x);  // row_half_mirror
  return x;
}

// Number of set bits of m below this lane (+ acc): v_mbcnt_lo / v_mbcnt_hi chain.
__device__ __forceinline__ int mbcnt(unsigned long long m, int acc = 0) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, acc));
}

// LDS layout of the lane-wise kernels (one trajectory per lane: closed-form models and the N <= 16 nets at 64 per wavefront),
// behind the MlpTile region when there is one.  Shared by the kernel and the host-side plan (ionode_capi.hip).
//   rows   [64][ROWB]  a lane's interpolant: {t0, step length} {1/step, 8 spare bytes} {5 x D fp64 coefficients}
//   aux    the fused objective's partial sums [64][8] fp64 (general and table variants; the lean variants sum no objective)
//   owp    [64] i32    the lane's protocol index
//   trl    [64] i32    the lane's trajectory index
//   clist  [512] u16   the attempt's dense-output WORK LIST: one entry per 8-sample chunk {lane, 8 * chunk number}
// gfx950 allocates LDS in 1280-byte granules; the lean 2-state kernels (<= 128 registers) want 16 wavefronts per compute unit:
// <= 10 240 bytes, the others 12: <= 12 800.
#ifndef IONODE_LEAN
#define IONODE_LEAN 1   // 0: A/B build without the contract folding of the TAIL 1 / 2 / & 8 variants
#endif
struct LwLds {
  // (D, tail): model states, the kernel's TAIL slot (0 general, 1 lean, 2 table).  Row stride (4 + 5 D) * 8 = 112 / 272 bytes:
  // consecutive rows start on different LDS banks (128-byte rows put every row on the same banks: measured 30 % of the LDS
  // cycles in bank conflicts).
  static __host__ __device__ constexpr int rowb(int D) { return (4 + 5 * D) * 8; }
  static __host__ __device__ constexpr int aux_off(int D) { return 64 * rowb(D); }
  static __host__ __device__ constexpr int aux_bytes(int tail) { return (tail == 1 && IONODE_LEAN) ? 0 : 64 * 64; }
  static __host__ __device__ constexpr int owp_off(int D, int tail) { return aux_off(D) + aux_bytes(tail); }
  static __host__ __device__ constexpr int trl_off(int D, int tail) { return owp_off(D, tail) + 256; }
  static __host__ __device__ constexpr int clist_off(int D, int tail) { return trl_off(D, tail) + 256; }
  static __host__ __device__ constexpr int bytes(int D, int tail) { return clist_off(D, tail) + 1024; }
};
static_assert(LwLds::bytes(2, 1) <= 10240 && LwLds::bytes(2, 0) <= 12800 && LwLds::bytes(2, 2) <= 12800, "2-state kernels: 16 / 12 wavefronts per compute unit");

#ifndef IONODE_PK_SAMPLES
#define IONODE_PK_SAMPLES 1
#endif

// Uniform protocol grid, in two halves so that a caller can issue the two sample loads of several lookups back to back:
// the sample index (false: t outside the protocol), and the interpolation from the two samples.
__device__ __forceinline__ bool protocol_index(const KArgs &a, double t, int &i);
__device__ __forceinline__ double protocol_from(const KArgs &a, double v_lo, double v_hi, int i, double t);

// interp1d(t, v) (linear) with the reference's out-of-range rule (train-s1.py:218-229, :234-237).
// scipy: i = searchsorted(x, t) [left], clipped to [1, n-1]; y = slope*(t - x[i-1]) + y[i-1].
// Uniform grids find i arithmetically (i = ceil((t - t0)/dt)); explicit grids by bisection.
__device__ __forceinline__ bool protocol_index(const KArgs &a, double t, int &i) {
  // branch-free (selects only): several lookups in a row become one block of arithmetic followed by one block of loads; the
  // index is valid (1 .. n-1) whatever t is, so the sample loads need no guard
  const int n = a.Np;
  const double t_last = a.prot_t0 + (double)(n - 1) * a.prot_dt;
  const bool inr = !(t < a.prot_t0 || t > t_last || t != t);
  const double u = div_by(t - a.prot_t0, a.prot_dt, a.prot_rdt);
  double ci = ceil(u);
  ci = (ci < 1.0) ? 1.0 : ci;
  ci = (ci > (double)(n - 1)) ? (double)(n - 1) : ci;
  i = inr ? (int)ci : 1;
  return inr;
}
__device__ __forceinline__ double protocol_from(const KArgs &a, double v_lo, double v_hi, int i, double t) {
  const double x_lo = a.prot_t0 + (double)(i - 1) * a.prot_dt;
  const double slope = div_by(v_hi - v_lo, a.prot_dt, a.prot_rdt);
  return slope * (t - x_lo) + v_lo;
}

__device__ __forceinline__ bool protocol_v(const KArgs &a, const double *__restrict__ pv, double t, double &v) {
  const int n = a.Np;
  if (a.prot_t != nullptr) {
    const double *__restrict__ x = a.prot_t;
    if (t < x[0] || t > x[n - 1] || t != t) { v = a.v_oob; return false; }
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (x[mid] < t) lo = mid + 1; else hi = mid;
    }
    const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const double slope = (pv[i] - pv[i - 1]) / (x[i] - x[i - 1]);
    v = slope * (t - x[i - 1]) + pv[i - 1];
    return true;
  }
  int i;
  if (!protocol_index(a, t, i)) { v = a.v_oob; return false; }
  v = protocol_from(a, pv[i - 1], pv[i], i, t);
  return true;
}

// nn.LeakyReLU(0.01): x > 0 ? x : 0.01*x  ==  max(x, 0.01*x) for every input (incl. +-0, NaN): 2 VALU ops
// fmaxf() makes hipcc canonicalise its operands first (v_max_f32 x, x, x: one dead vector instruction per MFMA result register -- 1280
// in the N <= 16 kernel at 64 per wavefront); the v_max_f32 instruction itself returns the same bits for every non-NaN input and quiets
// NaNs on its own (IEEE mode), so it is issued directly.  The multiply stays hipcc's: it is the first reader of the MFMA result and
// gets the required wait states; the asm reads its output, so it can only follow it.
__device__ __forceinline__ float lrelu(float x) {
  const float t = x * 0.01f;
  float h;
  asm("v_max_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(t));
  return h;
}

// ---------------------------------------------------------------------------------------------
// Stage MLP of one 16-trajectory tile on the fp32 MFMA.  All G wavefronts of the workgroup call
// eval() together in uniform control flow; it returns net([x0, x1]) for the lane's trajectory.
//
// Work split.  A hidden layer is NT row tiles x NT k-tiles of 16x16x4 MFMAs (4 per tile pair).  Each of the G
// wavefronts owns F = NT/G FULL row tiles (rt = w + i*G: all k-tiles) and a 1/G K-slice (k-tiles kt % G == w) of each
// of the R = NT - G*F REMAINDER row tiles (rt = G*F + j).  For N = 200 (NT = 13, G = 4) that is 3*13 + 13/4 tile
// products per wavefront instead of 4*13 on the critical wavefront.  The K-slices of a remainder tile are
// partial sums; they meet in LDS and every wavefront folds them itself after the layer barrier.
// Wavefront w walks the k-tiles in the rotated order kt = (s + w) mod NT, s = 0..NT-1, so that "this step carries my
// K-slice" is the compile-time predicate s % G == 0 (plus s + w < NT on the last such step) instead of a branch per
// k-step, and the weight stream of a wavefront has a static shape.
//
// Canonical accumulation order (the oracle executes exactly this; DESIGN.md "canonical MLP order"):
//   k index of (k-tile kt, k-step r, lane group q):  k = 16*kt + 4*q + r        (accumulator layout == B operand layout)
//   full tile rows:       acc = bias; for s = 0..NT-1, kt = (s + rt % G) mod NT: for r: for q: acc = fmaf(W[row][k], h[k], acc)
//   remainder tile rows:  p_w = (w == 0 ? bias : 0); for kt with kt % G == w, ascending: for r: for q: p_w = fmaf(...)
//                         acc = (p_0 + p_1) + (p_2 + p_3)                            (G = 4; G = 1 has no remainder)
//   Linear(N, 1):         part_q = 0; for kt: for r: part_q = fmaf(wl[k], h[k], part_q);
//                         out = ((part_0 + part_1) + (part_2 + part_3)) + bl
//
// Weight streaming.  The A fragments of the hidden layers are the only global traffic of the MLP.  Stream order
// (ionode_mlp_pack): layer | wavefront w | step s (k-tile (s + w) mod NT) | fragments | lane.  Every step has F
// fragments holding the full tiles k-step-major (element e = r*F + i -> float4 e/4, component e%4); steps with
// s % G == 0 carry R more, one per remainder tile (components = k-steps r; zeros when s + w >= NT).
// Fragments are consumed from a register ring
// refilled PD k-tiles ahead (PD == NT: a whole layer ahead) with SRSRC buffer loads issued right behind the last
// MFMA that reads them, pinned with sched_barrier so the machine scheduler neither sinks them to the end of the
// layer nor bunches them into an MFMA-free gap.  The stream runs at ~28 B/clk/CU; it is not the limiter (cutting its
// bytes by 17 % changed nothing): the MFMA count and the issue work per k-tile step and per layer boundary are.
// Small vectors (layer-0 rows, biases, last-layer weights) live in LDS for the kernel's lifetime.
// ---------------------------------------------------------------------------------------------
// Hand-scheduled hidden-layer stream (tools/gen_mlp_asm.py -> mlp_asm_nt13.inc): the N = 200 tile <4, 4, 13, 13> runs its
// hidden stack as ONE inline-asm statement with a fixed register map (weight ring in AGPRs a[0:171], working set in
// v[184:255]) and a software-pipelined layer boundary; same canonical accumulation order, same bits.  -DIONODE_NO_ASM_CORE
// builds the compiler-scheduled stream instead (A/B, stamps).
// Round 5: N = 200 pads its contraction index to 208; k-tile 12 holds eight real k and eight padding columns -- two of the four k of each of
// its MFMAs.  Without the padding terms (exact no-ops) the canonical chain through the tile is 192, 196, 193, 197 | 194, 198, 195, 199: the asm
// stream runs it as TWO MFMAs per accumulator (tools/gen_mlp_asm.py "short form"), and ionode_mlp_pack lays the tile's A fragments out for it.
#ifndef IONODE_KT12_SHORT
#if defined(IONODE_NO_ASM_CORE)
#define IONODE_KT12_SHORT 0
#else
#define IONODE_KT12_SHORT 1
#endif
#endif
#if !defined(IONODE_NO_ASM_CORE)
#define IONODE_ASM_CORE 1
#include "mlp_asm_nt13.inc"
#include "mlp_asm_nt13x2.inc"
#else
#define IONODE_ASM_CORE 0
#endif

template <int G, int RT, int NT, int PD, int TAIL = 0>
struct MlpTile {
  static constexpr bool ASM = IONODE_ASM_CORE && G == 4 && NT == 13 && PD == 13;
  static constexpr int GW = G;           // wavefronts per tile
  // TAIL == 4: TWO 16-trajectory column sets per tile (32 trajectories per workgroup; launches of >= 2 tiles per compute unit).
  // Every weight fragment then feeds two MFMAs, and wavefronts 0, 1 integrate set 0, wavefronts 2, 3 set 1: the scalar
  // Runge-Kutta work is replicated twice per trajectory instead of four times.  Asm stream only (tools/gen_mlp_asm.py --ns 2).
  static constexpr int NSETS = (TAIL == 4) ? 2 : 1;
  static_assert(NSETS == 1 || ASM, "two column sets exist for the asm tile only");
  static constexpr int F = NT / G;       // full row tiles per wavefront
  static constexpr int R = NT - G * F;   // remainder row tiles, K-split over the G wavefronts
  static constexpr int NP = 16 * NT;
  static constexpr int RP = (R > 0 ? R : 1);
  static constexpr int HT = NT + G - 1;  // activation slots per buffer (see Hs)
  static_assert(NT % PD == 0, "ring depth must divide the k-tile count");
  static_assert(TAIL == 0 || TAIL == 4, "tail scheme retired: the refills are interleaved with the MFMAs instead");
  static_assert(RT == F + R, "RT = full + remainder tile slots per wavefront");
  static_assert((4 * RT) % 4 == 0 && (RT == 1 || RT == 2 || RT == 4 || RT == 8), "fragment = RT float4 per k-tile");
  static_assert(R == 0 || G == 4, "the remainder combine tree is written for 4 wavefronts");
  static constexpr int NOWN = (NT + G - 1) / G;      // steps s = 0, G, 2G, ... carry a K-slice of the remainder tiles
  static constexpr int FRAGS = NT * F + NOWN * R;    // 1 KiB fragments per wavefront per layer
  // ring slot of step u (blocked scheme: step kt0 + u): F full fragments (+ R remainder fragments when owned)
  f32x4 ring[PD][F > 0 ? F : 1];
  // N = 100 (NT = 7: one full tile per wavefront + THREE remainder tiles): K-splitting three tiles over the wavefronts costs three
  // partial-sum exchanges and a 60-instruction fold per layer on every wavefront, and the layer barrier waits for it.  OWNREM:
  // wavefront w < R computes remainder tile w WHOLE -- its four canonical partial chains p_0..p_3 (k-tiles kt % 4 == c, ascending)
  // in four accumulators, folded in registers with the canonical tree -- so the layer's activations are complete at the barrier.
  // 56 / 56 / 56 / 28 MFMAs per layer instead of 52 / 52 / 52 / 40, no partial sums in LDS; same chains, same bits.  The chains
  // need the k-tiles in natural order (the full tile walks them rotated), so they read their own B operands, one step behind.
#ifndef IONODE_OWNREM
#define IONODE_OWNREM 1
#endif
  static constexpr bool OWNREM = IONODE_OWNREM && (G == 4 && F == 1 && R == 3 && PD == NT);
  f32x4 rrem[(R > 0 && !OWNREM) ? (PD + G - 1) / G : 1][(R > 0 && !OWNREM) ? RP : 1];
  f32x4 rown[OWNREM ? NT : 1];   // fragments of my remainder tile, one per k-tile (a layer ahead, like the ring)
  unsigned voff0;                // per lane: lane * 16 (fragments of another wavefront's stream: frag_of)
  // NT == 1 (N <= 16, architectures s03-s05): the whole hidden stack is LMAX fragments -- it stays in registers
  static constexpr bool TINY = (NT == 1 && G == 1);
  static constexpr int LMAX = 10;
  f32x4 wres[TINY ? LMAX : 1];
  // LDS [2][HT*64] activations after LeakyReLU, accumulator layout.  HT = NT + G - 1 slots: tiles 0..G-2 are stored
  // twice (slot kt and kt + NT) so that wavefront w reads its rotated sequence kt = (s + w) mod NT at the linear
  // address base_w + s -- an immediate offset, no per-step address arithmetic.  Remainder-tile slots are filled by
  // every wavefront itself (identical bits) when it folds the partial sums.
  f32x4 *Hs;
  f32x4 *Ps;          // LDS [2][R][G][64] partial sums of the remainder tiles (pre-activation)
  const f32x4 *W0s;   // LDS [NP] {b0, w00, w01, 0}
  const float *biasS; // LDS [L][NP]
  const float *wlS;   // LDS [NP] + bl
  __amdgpu_buffer_rsrc_t rsrc;  // weight image; one 32-bit VGPR offset per lane + scalar offset per load
  unsigned voff;      // per lane: byte offset of (this wavefront's stream, lane) inside a hidden layer
  unsigned hid0;      // byte offset of hidden layer 0 in the image
  unsigned lbytes;    // bytes per hidden layer in the image
  unsigned lds0;      // LDS byte address of the tile's region (asm stream)
  int bl_bits;        // bias of Linear(N, 1), wave-uniform (asm stream)
  int sw12;           // asm stream: the wavefront's index when k-tile 12 may take its two-MFMA form (N <= 200: k >= 200 is padding), else 99
  int L, wave, lane;
#ifdef IONODE_STAMPS
  Stamps *sp;
#define MSTAMP(slot) STAMP(*sp, slot)
#else
#define MSTAMP(slot) do { } while (0)
#endif

  static __host__ __device__ constexpr size_t layer_floats() { return (size_t)G * FRAGS * 256 + NP; }
  // index of step s's first fragment in a wavefront's layer stream
  static __host__ __device__ constexpr int step_base(int s) { return s * F + R * ((s + G - 1) / G); }
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return ((size_t)2 * NSETS * HT * 64 + (size_t)2 * NSETS * R * G * 64 + NP) * 16 + ((size_t)L * NP + NP + 4) * 4;
  }
  // the asm stream parks the stores of a not-yet-existing previous layer in a 1 KiB scratch slot behind the tile's LDS
  // (so that every pass issues the same LDS operations and the wait counts are static); two column sets: + 256 B through
  // which the wavefronts exchange their stage inputs
  static __host__ __device__ constexpr size_t scratch_off(int L) { return (lds_bytes(L) + 15) & ~(size_t)15; }
  static __host__ __device__ constexpr size_t lds_total(int L) { return ASM ? scratch_off(L) + 1024 + (NSETS > 1 ? 256 : 0) : lds_bytes(L); }

  __device__ __forceinline__ void init(const KArgs &a, unsigned char *smem, int wave_, int lane_, int first_traj = 0) {
    L = a.L; wave = wave_; lane = lane_;
    // the tile's weight image: the shared one, or image number first_traj / traj_per_img of an ensemble
    const float *__restrict__ img = a.mlp + (a.traj_per_img > 0 ? (size_t)(first_traj / a.traj_per_img) * (size_t)a.mlp_stride : (size_t)0);
    Hs = reinterpret_cast<f32x4 *>(smem);
    Ps = Hs + 2 * NSETS * HT * 64;
    f32x4 *w0 = Ps + 2 * NSETS * R * G * 64;
    float *bs = reinterpret_cast<float *>(w0 + NP);
    float *ws = bs + (size_t)L * NP;
    constexpr size_t lstride = layer_floats();
    const int tid = wave * 64 + lane;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(img);
    for (int i = tid; i < NP; i += 64 * G) w0[i] = src[i];
    for (int i = tid; i < L * NP; i += 64 * G) bs[i] = img[4 * (size_t)NP + (size_t)(i / NP) * lstride + (lstride - NP) + (i % NP)];
    const float *wl = img + 4 * (size_t)NP + (size_t)L * lstride;
    for (int i = tid; i < NP + 4; i += 64 * G) ws[i] = wl[i];
    W0s = w0; biasS = bs; wlS = ws;
    const size_t img_bytes = (4 * (size_t)NP + (size_t)L * lstride + NP + 4) * 4;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (int)img_bytes, 0x00020000);
    voff = (unsigned)(wave * FRAGS * 1024 + lane * 16);
    voff0 = (unsigned)(lane * 16);
    hid0 = (unsigned)(4 * NP * 4);
    lbytes = (unsigned)(lstride * 4);
    if constexpr (TINY) {
#pragma unroll
      for (int l = 0; l < LMAX; ++l) wres[l] = (l < L) ? frag(hid0 + (unsigned)l * lbytes, 0) : f32x4{0, 0, 0, 0};
    }
    if constexpr (ASM) {
#if IONODE_ASM_CORE
      lds0 = (unsigned)(uintptr_t)smem;
      if (L > 0)
        asm volatile(IONODE_MLPASM_INIT_13
                     :
                     : [voff] "v"(voff), [rsrc] "s"(rsrc), [hid0] "s"(hid0)
                     : "memory", "scc", IONODE_MLPASM_CLOBBER_A_13, IONODE_MLPASM_CLOBBER_S_13);
#endif
      __syncthreads();
      bl_bits = __builtin_amdgcn_readfirstlane(__float_as_int(wlS[NP]));
      sw12 = (IONODE_KT12_SHORT && a.N <= 200) ? wave : 99;
      return;
    }
    // prime the ring with the first PD steps of hidden layer 0
#pragma unroll
    for (int u = 0; u < (TINY ? 0 : PD); ++u) {
#pragma unroll
      for (int j = 0; j < F; ++j) ring[u][j] = frag(hid0, step_base(u) + j);
      if constexpr (R > 0 && !OWNREM) {
        if (u % G == 0) {
#pragma unroll
          for (int j = 0; j < R; ++j) rrem[u / G][j] = frag(hid0, step_base(u) + F + j);
        }
      }
    }
    if constexpr (OWNREM) {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) rown[kt] = frag_of(hid0, kt);
    }
    __syncthreads();
  }

  // OWNREM: the fragment of remainder tile `wave` for k-tile kt.  It sits in the stream of wavefront kt % G (the K-slice owner of
  // the packed layout, ionode_mlp_pack), at that wavefront's owned step kt - kt % G, behind the step's F full-tile fragments.
  __device__ __forceinline__ f32x4 frag_of(unsigned lbase, int kt) const {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned n = (unsigned)((kt % G) * FRAGS + step_base(kt - kt % G) + F) + (unsigned)(wave < R ? wave : 0);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff0, lbase + n * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
  }

  // one 1 KiB fragment (64 lanes x float4): fragment n of this wavefront's stream of the layer at byte offset `lbase`
  __device__ __forceinline__ f32x4 frag(unsigned lbase, int n) const {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, lbase + (unsigned)n * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
  }

  // activations of remainder tile j as a B operand / dot-product input: fold the G partial sums (fixed tree)
  __device__ __forceinline__ f32x4 remainder_h(const f32x4 *__restrict__ Pin, int j) const {
    const f32x4 p0 = Pin[(j * G + 0) * 64 + lane], p1 = Pin[(j * G + 1) * 64 + lane];
    const f32x4 p2 = Pin[(j * G + 2) * 64 + lane], p3 = Pin[(j * G + 3) * 64 + lane];
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = lrelu((p0[r] + p1[r]) + (p2[r] + p3[r]));
    return h;
  }

  // N <= 16: one wavefront, one 16x16 tile per layer, weights resident, activations never leave the registers
  // (the accumulator tile IS the next B operand).  Same canonical order as the general path with NT = 1.
  __device__ __forceinline__ float eval_tiny(float x0, float x1) {
    const int q = lane >> 4;
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 w = W0s[4 * q + r];
      h[r] = lrelu(fmaf(w[2], x1, fmaf(w[1], x0, w[0])));
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      if (l < L) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(biasS + l * NP + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[l][r], h[r], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu(acc[r]);
      }
    }
    const f32x4 w = *reinterpret_cast<const f32x4 *>(wlS + 4 * q);
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) part = fmaf(w[r], h[r], part);
    const float pair = part + __shfl_xor(part, 16);
    return (pair + __shfl_xor(pair, 32)) + wlS[NP];
  }

  // N <= 16 at 64 trajectories per wavefront (one per lane, no replicated scalar work): four 16-column tiles share the
  // resident weights.  Tile c holds trajectories 16c..16c+15; its layer-0 inputs are gathered from the owning lanes with
  // ds_bpermute, and its result for column n is the value of lane 16c + n.  Per tile this is eval_tiny() -- same canonical
  // order, same bits -- and the four tiles' MFMA chains are independent, so they fill each other's latency.
  __device__ __forceinline__ float eval_tiny64(float x0, float x1) {
    const int q = lane >> 4, n = lane & 15;
    f32x4 w0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w0[r] = W0s[4 * q + r];
    f32x4 h[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float a0 = __shfl(x0, 16 * c + n), a1 = __shfl(x1, 16 * c + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[c][r] = lrelu(fmaf(w0[r][2], a1, fmaf(w0[r][1], a0, w0[r][0])));
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      if (l < L) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(biasS + l * NP + 4 * q);
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = bias;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[l][r], h[c][r], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[c][r] = lrelu(acc[c][r]);
      }
    }
    const f32x4 w = *reinterpret_cast<const f32x4 *>(wlS + 4 * q);
    const float bl = wlS[NP];
    float res = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float part = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(w[r], h[c][r], part);
      const float pair = part + __shfl_xor(part, 16);
      const float out = (pair + __shfl_xor(pair, 32)) + bl;
      if (c == q) res = out;
    }
    return res;
  }

  __device__ __forceinline__ float eval(float x0, float x1) {
    if constexpr (TINY) return eval_tiny(x0, x1);
    const int q = lane >> 4;
    constexpr int tstride = HT * 64;
    constexpr int pstride = R * G * 64;
    MSTAMP(0);  // slot 0: everything outside the MLP (RK scalar work, emission)
#if IONODE_ASM_CORE
    if constexpr (ASM) {
      if (L > 0) {
        // the whole evaluation -- Linear(2, N), the hidden stack, Linear(N, 1) -- is one asm statement (tools/gen_mlp_asm.py).
        // Inputs: per-lane LDS byte addresses of the two activation buffers (b = 0: input of even layers), the partial-sum
        // buffers, this lane's rows of the small vectors, and the wavefront's weight stream.
        constexpr unsigned HB = (unsigned)(NSETS * tstride) * 16u, PB = (unsigned)(NSETS * pstride) * 16u;  // bytes per activation / partial-sum buffer
        const unsigned hw0 = lds0 + (unsigned)(wave * 64 + lane) * 16u, hw1 = hw0 + HB;
        const unsigned fw0 = lds0 + (unsigned)((NT - 1) * 64 + lane) * 16u, fw1 = fw0 + HB;
        const unsigned pl0 = lds0 + 2u * HB + (unsigned)lane * 16u, pl1 = pl0 + PB;
        const unsigned pw0 = pl0 + (unsigned)wave * 1024u, pw1 = pl1 + (unsigned)wave * 1024u;
        const unsigned bias0 = (unsigned)(uintptr_t)biasS, w00 = (unsigned)(uintptr_t)W0s;
        const unsigned bias_a = bias0 + (unsigned)(16 * wave + 4 * q) * 4u, bias_r = bias0 + (unsigned)(16 * (NT - 1) + 4 * q) * 4u;
        const unsigned w0a = w00 + (unsigned)(16 * wave + 4 * q) * 16u, w0r = w00 + (unsigned)(16 * (NT - 1) + 4 * q) * 16u;
        const unsigned wla = (unsigned)(uintptr_t)wlS + (unsigned)q * 16u;
        const unsigned dummy = lds0 + (unsigned)scratch_off(L) + (unsigned)lane * 16u;
        const int nl = __builtin_amdgcn_readfirstlane(L);
        float out;
        if constexpr (NSETS == 1) {
          asm volatile(IONODE_MLPASM_LAYERS_13
                       : [out] "=v"(out)
                       : [hw_in] "v"(hw0), [hw_out] "v"(hw1), [fw_in] "v"(fw0), [fw_out] "v"(fw1), [pl_in] "v"(pl0), [pl_out] "v"(pl1),
                         [pw_in] "v"(pw0), [pw_out] "v"(pw1), [bias_a] "v"(bias_a), [bias_r] "v"(bias_r), [voff] "v"(voff),
                         [dummy] "v"(dummy), [w0a] "v"(w0a), [w0r] "v"(w0r), [wla] "v"(wla), [x0] "v"(x0), [x1] "v"(x1),
                         [rsrc] "s"(rsrc), [nl] "s"(nl), [lbytes] "s"(lbytes), [hid0] "s"(hid0), [wave] "s"(wave), [bl] "s"(bl_bits), [sw] "s"(sw12)
                       : "memory", "scc", "vcc", IONODE_MLPASM_CLOBBER_V_13, IONODE_MLPASM_CLOBBER_A_13, IONODE_MLPASM_CLOBBER_S_13);
        } else {
          // two column sets: this wavefront's stage inputs belong to set `wave / 2`; the stream exchanges them through LDS
          // ([set][16] x {x0, x1} behind the scratch slot) and returns the result of the own set
          const int cset = wave / (G / NSETS);
          const unsigned xch = lds0 + (unsigned)scratch_off(L) + 1024u + (unsigned)(lane & 15) * 8u;
          const unsigned xchw = xch + (unsigned)cset * 128u;
          const int own_h = cset * (int)(tstride * 16), own_p = cset * (int)(pstride * 16);
          asm volatile(IONODE_MLPASM_LAYERS_13x2
                       : [out] "=v"(out)
                       : [hw_in] "v"(hw0), [hw_out] "v"(hw1), [fw_in] "v"(fw0), [fw_out] "v"(fw1), [pl_in] "v"(pl0), [pl_out] "v"(pl1),
                         [pw_in] "v"(pw0), [pw_out] "v"(pw1), [bias_a] "v"(bias_a), [bias_r] "v"(bias_r), [voff] "v"(voff),
                         [dummy] "v"(dummy), [w0a] "v"(w0a), [w0r] "v"(w0r), [wla] "v"(wla), [x0] "v"(x0), [x1] "v"(x1),
                         [xchw] "v"(xchw), [xchr] "v"(xch),
                         [rsrc] "s"(rsrc), [nl] "s"(nl), [lbytes] "s"(lbytes), [hid0] "s"(hid0), [wave] "s"(wave), [bl] "s"(bl_bits), [sw] "s"(sw12),
                         [own_h] "s"(own_h), [own_p] "s"(own_p)
                       : "memory", "scc", "vcc", IONODE_MLPASM_CLOBBER_V_13x2, IONODE_MLPASM_CLOBBER_A_13x2, IONODE_MLPASM_CLOBBER_S_13x2);
        }
        MSTAMP(3);  // slot 3: the whole evaluation (asm stream)
        return out;
      }
    }
#endif

    // layer 0: Linear(2, N) + LeakyReLU on the VALU, written in accumulator layout; row tile rt by wavefront rt % G.
    // hOwn = this wavefront's tile `wave`: the k-tile it consumes at step 0 of the next layer ((0 + w) mod NT).
    // That B operand is taken from registers, which lets the layer barrier sit AFTER step 0: the LDS store -> barrier ->
    // load round trip of the activations and the wait for the slowest wavefront overlap with step 0's MFMAs.
    f32x4 hOwn = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int rt = wave + i * G;
      if (rt < NT) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 w = W0s[16 * rt + 4 * q + r];
          h[r] = lrelu(fmaf(w[2], x1, fmaf(w[1], x0, w[0])));
        }
        if (i == 0) hOwn = h;
        Hs[rt * 64 + lane] = h;
        if (rt < G - 1) Hs[(rt + NT) * 64 + lane] = h;
      }
    }
    MSTAMP(1);  // slot 1: layer 0

    if constexpr (!ASM)  // (the asm tile comes here only with L == 0)
    for (int l = 0; l < L; ++l) {
      f32x4 *__restrict__ Hin = Hs + (l & 1) * tstride;
      f32x4 *__restrict__ Hout = Hs + ((l + 1) & 1) * tstride;
      const f32x4 *__restrict__ Pin = Ps + (l & 1) * pstride;
      f32x4 *__restrict__ Pout = Ps + ((l + 1) & 1) * pstride;
      const int ln = (l + 1 < L) ? l + 1 : 0;  // the ring runs cyclically over the hidden stack
      // (L == 1 simply re-streams the same layer: a runtime 'resident' branch around the refills would make
      // hipcc's wait-count pass lose the age of the loads and drain them all at every use)
      const unsigned lcur = hid0 + (unsigned)l * lbytes, lnext = hid0 + (unsigned)ln * lbytes;

      // B operands of the remainder k-tiles: layer 0 wrote them as ordinary tiles; hidden layers leave partial sums,
      // which every wavefront folds into the remainder slots of the input buffer itself (identical bits from all
      // wavefronts; each reads after its own write, so no barrier).  The fold is needed first at step G*F - wave;
      // when that is late enough it runs behind the MFMAs of step 0 instead of in the layer prologue.
      constexpr bool LAZY_FOLD = (R > 0) && !OWNREM && (G * F - (G - 1) >= 3);
      f32x4 acc[F > 0 ? F : 1], accr[RP];
#pragma unroll
      for (int i = 0; i < F; ++i)
        acc[i] = *reinterpret_cast<const f32x4 *>(biasS + l * NP + 16 * (wave + i * G) + 4 * q);
      f32x4 pc[OWNREM ? 4 : 1];   // OWNREM: the four partial chains of my remainder tile
      f32x4 bn_nxt = f32x4{0, 0, 0, 0};
      if constexpr (OWNREM) {
        pc[0] = *reinterpret_cast<const f32x4 *>(biasS + l * NP + 16 * (G * F + (wave < R ? wave : 0)) + 4 * q);  // chain 0 carries the bias
        pc[1] = pc[2] = pc[3] = f32x4{0, 0, 0, 0};
      } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const f32x4 bz = *reinterpret_cast<const f32x4 *>(biasS + l * NP + 16 * (G * F + j) + 4 * q);
        accr[j] = (wave == 0) ? bz : f32x4{0, 0, 0, 0};  // partial sum 0 carries the bias
      }
      }
      for (int kt0 = 0; kt0 < NT; kt0 += PD) {
        static_assert(R == 0 || PD == NT, "remainder tiles need the full-layer ring (static step index)");
        const bool same_layer = kt0 + PD < NT;
        // stream position of the refills issued in this block: same layer, PD steps ahead, or the next layer's start
        const unsigned lref = same_layer ? lcur + (unsigned)step_base(kt0 + PD) * 1024u : lnext;
        // this wavefront's k-tile at step kt0 + u is (kt0 + u + wave) mod NT = slot kt0 + u + wave of the buffer
        const f32x4 *__restrict__ Bw = Hin + (kt0 + wave) * 64 + lane;
        f32x4 b_nxt = hOwn;                      // step 0 of the layer: own tile, from registers (before the barrier)
        if (kt0 > 0) b_nxt = Bw[0];
        MSTAMP(8);  // slot 8: layer prologue (bias)
#pragma unroll
        for (int u = 0; u < PD; ++u) {
          if (u == 1) MSTAMP(9);       // slot 9: first k-tile (+ barrier)
          if (u == PD - 1) MSTAMP(3);  // slot 3: k-tiles 1..PD-2
          const f32x4 b = b_nxt;
#ifndef IONODE_EXPERIMENT_NO_BREAD  // timing experiment only
          // LDS read one step ahead, immediate offset -- except across the layer barrier (after step 0)
          if (u + 1 < PD && !(u == 0 && kt0 == 0)) b_nxt = Bw[(u + 1) * 64];
#endif
          // K-slice ownership: static, except that the last owned step wraps past NT for the higher wavefronts
          const bool own_static = (R > 0) && !OWNREM && (u % G == 0);
          // OWNREM: my remainder tile's chain (u - 1) % 4 takes k-tile u - 1 (natural order, one step behind: k-tile 0 is another
          // wavefront's tile and exists only behind the layer barrier, which sits at the end of step 0)
          f32x4 bn = bn_nxt;
          if constexpr (OWNREM) {
            if (u >= 1) bn_nxt = Hin[u * 64 + lane];
          }
          const bool own_always = own_static && (u + G - 1 < NT);
          const bool own = own_static && (own_always || (u + wave < NT));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < F; ++i) {
              const int e = r * F + i;
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][e / 4][e % 4], b[r], acc[i], 0, 0, 0);
            }
            if constexpr (!OWNREM) {
            if (own_static) {
              if (own) {
#pragma unroll
                for (int j = 0; j < R; ++j)
                  accr[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(rrem[u / G][j][r], b[r], accr[j], 0, 0, 0);
              }
            }
            } else {
              if (u >= 1) {   // (every wavefront, also the one without a remainder tile: a wave-dependent branch around the
                              // refills makes hipcc drain every load at every use -- 29.8 -> 48.8 ms; it computes tile 0 again, unused)
                pc[(u - 1) % G] = __builtin_amdgcn_mfma_f32_16x16x4f32(rown[u - 1][r], bn[r], pc[(u - 1) % G], 0, 0, 0);
                if (r == 3) rown[u - 1] = frag_of(lref, u - 1);
              }
            }
            // refill every fragment whose last reader was this k-step; pinned here (see header comment)
#ifndef IONODE_EXPERIMENT_NO_REFILL  // timing experiment only: results are wrong for L > 1
#pragma unroll
            for (int j = 0; j < F; ++j)
              if ((4 * j + 3) / F == r) ring[u][j] = frag(lref, step_base(u) + j);
            if constexpr (!OWNREM) {
            if (own_static && r == 3) {
#pragma unroll
              for (int j = 0; j < R; ++j) rrem[u / G][j] = frag(lref, step_base(u) + F + j);
            }
            }
#endif
            if (LAZY_FOLD && u == 1 && r == 0 && l > 0) {
#pragma unroll
              for (int j = 0; j < R; ++j) Hin[(G * F + j) * 64 + lane] = remainder_h(Pin, j);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (u == 0 && kt0 == 0) {
            // ---- the layer barrier: everybody's activations / partial sums of the previous layer are in LDS ----
            if (G > 1) __syncthreads();
            if (R > 0 && !OWNREM && !LAZY_FOLD && l > 0) {
#pragma unroll
              for (int j = 0; j < R; ++j) Hin[(G * F + j) * 64 + lane] = remainder_h(Pin, j);
            }
            if (PD > 1) b_nxt = Bw[64];
            if constexpr (OWNREM) {
              bn_nxt = Hin[lane];   // k-tile 0 for my remainder chains (step 1)
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if constexpr (OWNREM) {
        const f32x4 bn = bn_nxt;   // k-tile NT - 1
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[(NT - 1) % G] = __builtin_amdgcn_mfma_f32_16x16x4f32(rown[NT - 1][r], bn[r], pc[(NT - 1) % G], 0, 0, 0);
        rown[NT - 1] = frag_of(lnext, NT - 1);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu((pc[0][r] + pc[1][r]) + (pc[2][r] + pc[3][r]));  // the canonical combine tree
        if (wave < R) Hout[(G * F + wave) * 64 + lane] = h;
      }
      MSTAMP(10);  // slot 10: last k-tile
#pragma unroll
      for (int i = 0; i < F; ++i) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu(acc[i][r]);
        if (i == 0) hOwn = h;
        Hout[(wave + i * G) * 64 + lane] = h;
        if (wave + i * G < G - 1) Hout[(wave + i * G + NT) * 64 + lane] = h;
      }
      if constexpr (!OWNREM) {
#pragma unroll
      for (int j = 0; j < R; ++j) Pout[(j * G + wave) * 64 + lane] = accr[j];
      }
      MSTAMP(4);  // slot 4: LeakyReLU + activation store
    }
    if (G > 1) __syncthreads();  // the last hidden layer's (or layer 0's) activations for the output layer
    MSTAMP(2);

    // Linear(N, 1) on the VALU: four partial fmaf chains (one per lane group q), fixed combine tree
    const f32x4 *__restrict__ Hin = Hs + (L & 1) * tstride;
    const f32x4 *__restrict__ Pin = Ps + (L & 1) * pstride;
    float part = 0.0f;
    // (issuing the LDS reads of several steps together was measured: no change at N = 200 / 100, -2 % at N = 500 -- the four wavefronts
    // run this chain redundantly and its stalls overlap; the 4-trajectory tile, one chain per evaluation on the critical path, does it)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f32x4 w = *reinterpret_cast<const f32x4 *>(wlS + 16 * kt + 4 * q);
      f32x4 h;
      if (R > 0 && !OWNREM && kt >= G * F && L > 0) h = remainder_h(Pin, kt - G * F);
      else h = Hin[kt * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(w[r], h[r], part);
    }
    const float pair = part + __shfl_xor(part, 16);   // (p0 + p1) or (p2 + p3)
    const float out = (pair + __shfl_xor(pair, 32)) + wlS[NP];
    if (G > 1 && (L & 1) == 0) __syncthreads();  // next evaluation's layer 0 rewrites buffer 0
    MSTAMP(5);  // slot 5: last layer
    return out;
  }
};

// ---------------------------------------------------------------------------------------------
// N = 200 at FOUR trajectories per tile: the small-batch / single-call form (round 4).  The reference's own scripts call
// odeint with ONE trajectory (train-s1.py:319-330, 32 sequential solves at :566-580), and BASELINE configs[4]'s per-GPU share is 1024
// trajectories: at 16 per tile those occupy 64 of 256 compute units and every RHS evaluation still costs a full 16-column MFMA
// pass (36.6 k cycles).  Here a tile is 4 trajectories (256 tiles for 1024 trajectories: the whole chip) and a hidden layer is
// 13 x 16 v_mfma_f32_4x4x1_16B_f32 per wavefront: 16 blocks of (4 rows) x (4 trajectories) x (1 k) -- each block an exact fmaf
// per element, so a block's accumulator runs the SAME chain as a row of the 16-column tile when it is fed the same k sequence.
//   lane = 4 b + i supplies A = W[row(b, i)][k];  lane = 4 b + j supplies B = h[k][trajectory j];  D[i][j] = VGPR i of lane 4 b + j
//   wavefront w, block b = 4 g + u:  g < 3: full row tile w + 4 g, rows 16 (w + 4 g) + 4 u + i -- all three tiles have rt % 4 == w, so
//                                    the whole wavefront walks the k-tiles in ONE rotated order kt = (s + w) mod 13, s = 0 .. 12;
//                                    g == 3: remainder tile 12, rows 192 + 4 u + i, partial chain w (k-tiles kt % 4 == w, ascending):
//                                    exactly the steps s % 4 == 0 with s + w < 13 of that same walk; on the other steps its A operand
//                                    is -0.0f (x + (-0 * h) == x for every x as long as h is finite and the chain is not at -0)
//   within a k-tile:  for r: for q: k = 16 kt + 4 q + r      (the canonical order; one MFMA per (r, q))
// so results are bit-identical to MlpTile<4, 4, 13, 13> and to the oracle.  Activations live in LDS as [k / 4][trajectory] float4
// (a lane reads the 4 x float4 of a k-tile for ITS trajectory; an output block IS one such float4); the remainder tile's four partial
// sums meet in LDS and every lane folds them itself ((p0 + p1) + (p2 + p3), LeakyReLU) when the walk reaches k-tile 12.
// Weights stream from L2 as in the 16-column tile (SRSRC buffer loads into a register ring one layer ahead: 13 steps x 4 float4),
// in their own image section (ionode_mlp_pack): layer | wavefront | step | q | lane -> float4 over r, then the layer's bias float4s.
// ---------------------------------------------------------------------------------------------
struct MlpTile4 {
  static constexpr int GW = 4, NT = 13, NP = 208;
  static constexpr int SLOTS = NT + 3;               // k-tile slots per activation buffer: tiles 0..2 are stored twice (slot kt and kt + 13), so
                                                     // that a block whose row tile rotates from k-tile g reads its walk kt = (s + g) mod 13 at the LINEAR slot s + g
  static constexpr int ACT = SLOTS * 16;             // float4 per activation buffer: [slot][q][trajectory]
  // Round 5: the lane layout of the one-trajectory tile (MlpRow1).  Wavefronts 0..2 hold 64 FULL rows each -- 16 blocks of 4 rows: block
  // b = 4 g + u is rows 16 (4 w + g) + 4 u + i, whose canonical chain rotates from k-tile g (the B operand is read per lane, so the four
  // block groups of a wavefront walk four rotations) -- and wavefront 3 holds the four partial chains of the sixteen remainder rows (block
  // 4 c + u: chain c of rows 192 + 4 u + i; 4 steps instead of 13), folded (p0 + p1) + (p2 + p3) across its lane groups.  172 one-KiB weight
  // loads per layer instead of 208 (round 4: every wavefront 48 rows + a remainder chain): this tile's walk waits on the compute unit's
  // vector-memory path as much as on the 4x4x1 MFMA's dependent issue (without its refills an evaluation takes 8.4 instead of 10.2 us).
  static constexpr int FRAGS_FULL = NT * 4, FRAGS_REM = 4 * 4;   // 1 KiB fragments per layer of a full-row wavefront / of the remainder wavefront
  static constexpr size_t layer_floats() { return (size_t)(3 * FRAGS_FULL + FRAGS_REM) * 256 + (size_t)4 * 256; }   // fragments + accumulator-start float4 per (wave, lane)
  static __host__ __device__ constexpr size_t lds_bytes(int L) {
    return ((size_t)2 * ACT + NP) * 16 + ((size_t)NP + 4) * 4 + (size_t)L * 64 * 16;   // activations x2, W0 rows, wl + bl, accumulator starts [L][wave][block]
  }
  f32x4 ring[NT][4];
  f32x4 w0r[4];    // layer 0: the four rows {b0, w00, w01, 0} of this lane's output block
                   // (round 4 also kept the lane's chain of the output weights resident: 52 registers the two walk forms of round 5 need;
                   // they are read from LDS together with the thirteen activation reads of the output layer -- one round trip)
  f32x4 *Hs;
  const f32x4 *W0s, *B4s;
  const float *wlS;
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, sec0, lbytes;
  int L, wave, lane;
#ifdef IONODE_STAMPS
  Stamps *sp;
#endif
  // offset (floats) of the T4 section inside the packed image of (L, N = 200): behind the 16-column image
  static __host__ __device__ constexpr size_t section_off(int L) {
    return 4 * (size_t)NP + (size_t)L * ((size_t)4 * 43 * 256 + NP) + NP + 4;   // MlpTile<4, 4, 13, 13>: FRAGS = 13 * 3 + 4 = 43 per wavefront
  }
  __device__ __forceinline__ f32x4 frag(unsigned lbase, int n) const {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, lbase + (unsigned)n * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
  }
  __device__ __forceinline__ void init(const KArgs &a, unsigned char *smem, int wave_, int lane_, int first_traj = 0) {
    L = a.L; wave = wave_; lane = lane_;
    const float *__restrict__ img = a.mlp + (a.traj_per_img > 0 ? (size_t)(first_traj / a.traj_per_img) * (size_t)a.mlp_stride : (size_t)0);
    Hs = reinterpret_cast<f32x4 *>(smem);
    f32x4 *w0 = Hs + 2 * ACT;
    float *ws = reinterpret_cast<float *>(w0 + NP);
    const int tid = wave * 64 + lane;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(img);
    for (int i = tid; i < NP; i += 256) w0[i] = src[i];
    const float *wl = img + 4 * (size_t)NP + (size_t)L * ((size_t)4 * 43 * 256 + NP);
    for (int i = tid; i < NP + 4; i += 256) ws[i] = wl[i];
    W0s = w0; wlS = ws;
    // accumulator starts of every layer and block (the four lanes of a block share them) into LDS: fetched from the image at the start
    // of a layer they would cost an L2 round trip per layer on the critical path of a single trajectory
    f32x4 *b4 = reinterpret_cast<f32x4 *>(ws + NP + 4);
    const size_t sec = section_off(L);
    for (int i = tid; i < L * 64; i += 256) {
      const int l = i >> 6, wv = (i >> 4) & 3, bb = i & 15;
      b4[i] = *reinterpret_cast<const f32x4 *>(img + sec + (size_t)l * layer_floats() + (size_t)(3 * FRAGS_FULL + FRAGS_REM) * 256 + (size_t)wv * 256 + (size_t)bb * 16);
    }
    B4s = b4;
    const size_t img_bytes = (sec + (size_t)L * layer_floats()) * 4;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (int)img_bytes, 0x00020000);
    sec0 = (unsigned)(sec * 4);
    lbytes = (unsigned)(layer_floats() * 4);
    voff = (unsigned)(wave * FRAGS_FULL * 1024 + lane * 16);
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) ring[s][q] = (L > 0 && (wave < 3 || s < 4)) ? frag(sec0, s * 4 + q) : f32x4{0, 0, 0, 0};
    {
      const int b = lane >> 2, kq0 = 16 * wave + b;
#pragma unroll
      for (int r = 0; r < 4; ++r) w0r[r] = (kq0 < NP / 4) ? src[4 * kq0 + r] : f32x4{0, 0, 0, 0};
    }
    __syncthreads();
  }
  // store an output block (4 rows of k-tile kt, lane group q, trajectory j); tiles 0..2 also at their second slot
  __device__ __forceinline__ void put_h(f32x4 *__restrict__ H, int kt, int q, int j, f32x4 h) const {
    H[(kt * 4 + q) * 4 + j] = h;
    if (kt < 3) H[((kt + NT) * 4 + q) * 4 + j] = h;
  }
  // The steps of this lane's block: step s reads the four float4 {h[16 kt + 4 q + r]}_r of the lane's trajectory from slot (slot0 + s * stride)
  // and runs the sixteen MFMAs of the k-tile in the canonical order (r-major, q-minor); the ring's fragments of the step are refilled for the
  // coming layer right behind their last use.  ONE code path for both kinds of wavefront (two instantiations merged the 208-register ring
  // through a branch and spilled): the remainder wavefront leaves after its four steps (a wave-uniform exit), its slot stride is a run-time value.
  __device__ __forceinline__ void walk(f32x4 &acc, const f32x4 *__restrict__ Bw, int sstride, int nsteps, unsigned lnext) {
    f32x4 hn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hn[q] = Bw[q * 4];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
      if (s == 4 && nsteps == 4) break;
      f32x4 hq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) hq[q] = hn[q];
      if (s + 1 < NT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hn[q] = Bw[(s + 1) * sstride + q * 4];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(ring[s][q][r], hq[q][r], acc, 0, 0, 0);
#ifndef IONODE_T4_NOREFILL   // timing experiment only (wrong results for L > 1): the walk without its weight stream
        if (r == 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) ring[s][q] = frag(lnext, s * 4 + q);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __device__ __forceinline__ float eval(float x0, float x1) {
    const int j = lane & 3, b = lane >> 2, g = b >> 2, u = b & 3;
    MSTAMP(0);  // slot 0: everything outside the MLP
    // layer 0: Linear(2, N) + LeakyReLU; the lane fills output block (kt, q) = (4 wave + g, u) = kq / 4, kq % 4 of its trajectory
    {
      const int kq = 16 * wave + b;
      if (kq < NP / 4) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu(fmaf(w0r[r][2], x1, fmaf(w0r[r][1], x0, w0r[r][0])));
        put_h(Hs, kq >> 2, kq & 3, j, h);
      }
    }
    f32x4 acc_next = (L > 0) ? B4s[wave * 16 + b] : f32x4{0, 0, 0, 0};
    __syncthreads();
    MSTAMP(1);  // slot 1: layer 0 + barrier
    for (int l = 0; l < L; ++l) {
      const f32x4 *__restrict__ Hin = Hs + (l & 1) * ACT;
      f32x4 *__restrict__ Hout = Hs + ((l + 1) & 1) * ACT;
      const int ln = (l + 1 < L) ? l + 1 : 0;   // the ring runs cyclically over the hidden stack (see MlpTile)
      const unsigned lnext = sec0 + (unsigned)ln * lbytes;
      // accumulators: D[i][j] = VGPR i: bias of row i of my block (partial chains c > 0 of the remainder rows start at 0: the image says so);
      // read one layer ahead
      f32x4 acc = acc_next;
      if (l + 1 < L) acc_next = B4s[((l + 1) * 4 + wave) * 16 + b];
      MSTAMP(2);  // slot 2: layer prologue
      // full rows (wavefronts 0..2): block group g walks k-tile (s + g) mod 13 = slot s + g, 13 steps; remainder rows (wavefront 3): block
      // group c = g runs partial chain c over the k-tiles c, c + 4, c + 8 (, 12: chain 0 only -- the others' step 3 reads the duplicate slots
      // 13..15 against -0.0f weights): 4 steps, 4 slots apart
      walk(acc, Hin + (g * 4) * 4 + j, (wave < 3) ? 16 : 64, (wave < 3) ? NT : 4, lnext);
      MSTAMP(3);  // slot 3: the MFMA walk
      if (wave < 3) {
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu(acc[r]);
        put_h(Hout, 4 * wave + g, u, j, h);
      } else {
        // the four chains of a row meet across the lane groups: (p0 + p1) + (p2 + p3)
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pair = acc[r] + __shfl_xor(acc[r], 16);
          h[r] = lrelu(pair + __shfl_xor(pair, 32));
        }
        if (lane < 16) put_h(Hout, NT - 1, u, j, h);
      }
      __syncthreads();
      MSTAMP(4);  // slot 4: LeakyReLU + store + layer barrier
    }
    // Linear(N, 1): chain q = b & 3 per lane (k = 16 kt + 4 q + r, kt ascending, r ascending), folded ((p0 + p1) + (p2 + p3)) + bl
    const f32x4 *__restrict__ Hin = Hs + (L & 1) * ACT;
    const int q = b & 3;
    // all thirteen activation reads in flight at once: one LDS round trip instead of thirteen
    f32x4 hl[NT], wv[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      hl[kt] = Hin[(kt * 4 + q) * 4 + j];
      wv[kt] = *reinterpret_cast<const f32x4 *>(wlS + 16 * kt + 4 * q);
    }
    float part = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(wv[kt][r], hl[kt][r], part);
    }
    const float pair = part + __shfl_xor(part, 4);    // (p0 + p1) or (p2 + p3): lanes 4 apart hold neighbouring chains
    const float out = (pair + __shfl_xor(pair, 8)) + wlS[NP];
    if ((L & 1) == 0) __syncthreads();   // the next evaluation's layer 0 rewrites buffer 0, which an even stack's last layer reads (odd: buffer 1)
    MSTAMP(5);  // slot 5: Linear(N, 1) + closing barrier
    return out;
  }
};

// ---------------------------------------------------------------------------------------------
// N = 200 at ONE trajectory per tile (round 5): the form for the reference's own call shape -- odeint(func, y0, t) with y0 of shape
// (1, 2) at all 155 call sites (train-s1.py:319-330; 32 sequential solves at :566-580).  One trajectory's evaluations are a serial
// chain; what bounds a chain link is how fast ONE compute unit can run a 208 x 208 matrix-vector product five times.  On the 4-trajectory
// tile that is the dependent-issue rate of v_mfma_f32_4x4x1 (17 cycles per k) for three idle columns out of four, and 208 weight loads
// per layer through the compute unit's one vector-memory path.  Here a LANE owns a ROW: acc = fmaf(W[row][k], h[k], acc) is one
// v_fmac_f32 per k (9 cycles dependent, tools/ubench/valu_chain.hip), with
//   * the weight W[row][k] in the lane's own register (streamed from L2 into a ring one layer ahead, as in the other tiles), and
//   * the activation h[k] broadcast by DPP: a lane holds {h[16 kt + 4 q + (lane & 3)]}, q = 0..3, of its current k-tile (ONE 16-byte LDS read
//     per lane and k-tile from a buffer kept in that transposed order), and v_fmac_f32_dpp quad_perm:[r, r, r, r] hands every lane of a
//     quad the value of its lane r: k = 16 kt + 4 q + r -- no cross-lane instruction, no SGPR traffic.  The LDS address is per lane, so the
//     four 16-lane groups of a wavefront can walk the k-tiles in four different rotations.
// Lanes: wavefronts 0..2, lane l: row 64 w + l = row tile 4 w + g (g = l / 16), whose canonical chain walks kt = (s + g) mod 13 -- every lane of
// these wavefronts owns a full row; wavefront 3, lane 16 c + i: partial chain c of remainder row 192 + i (k-tiles c, c + 4, c + 8 (, 12): FOUR
// steps instead of thirteen), folded (p0 + p1) + (p2 + p3) inside the wavefront.  172 weight loads per layer instead of 208: the vector-memory
// path (64 B/clk per compute unit whatever the lanes carry: masking idle lanes or sending them out of range changed nothing) is what bounds
// this tile.  Same canonical chains as every other form: bit-identical to the 16- and 4-trajectory tiles and to the oracle.
// Linear(N, 1): chain q on the lanes with (lane & 3) == q (k = 16 kt + 4 q + r, kt and r ascending) from a copy of the activations in
// natural order, folded by two DPP quad permutes.
// Image section (ionode_mlp_pack, behind the 4-trajectory tile's): per layer: wavefronts 0..2: [w][step s][r][lane] float4 over q of
// W[64 w + lane][16 ((s + lane / 16) mod 13) + 4 q + r]; wavefront 3: [step j][r][lane = 16 c + i] float4 over q of W[192 + i][16 (c + 4 j) + 4 q + r]
// (-0.0f for c + 4 j > 12); then per (wavefront, lane) the accumulator start (the row's bias; chains c > 0: 0).
// ---------------------------------------------------------------------------------------------
template <int MRES_> struct MlpRow1T {
  static constexpr int GW = 4, NT = 13, NP = 208;
  static constexpr int SLOTS = NT + 3;       // activation buffer, transposed order: k-tile slots 0..15, tiles 0..2 stored twice (slot kt and kt + 13)
  static constexpr int FRAGS_FULL = NT * 4, FRAGS_REM = 4 * 4;   // 1 KiB fragments per layer of a full-row wavefront / of the remainder wavefront
  static __host__ __device__ constexpr size_t layer_floats() { return (size_t)(3 * FRAGS_FULL + FRAGS_REM) * 256 + 256; }
  // floats: activations x2 (transposed) + natural copy x2 + accumulator starts [L][256] + wl[208] + bl(4)
  static __host__ __device__ constexpr size_t small_bytes(int L) { return ((size_t)2 * SLOTS * 16 + 2 * NP + (size_t)L * 256 + NP + 4) * 4; }
  // LDS-RESIDENT WEIGHTS: the compute unit's vector-memory path (64 B/clk) is what bounds this tile, and the workgroup has the whole 160 KB of
  // LDS to itself: the fragments of the first MRES = 2 steps of EVERY hidden layer of the three full-row wavefronts (24 KB per layer) stay in
  // LDS for the kernel's lifetime; their ring slots are refilled from there instead of from L2 -- 24 of 172 loads per layer less through the
  // memory path.  The same code for every layer (no per-layer variant: a branch around refills costs hipcc's wait counts their precision).
  // Stacks of more than 6 hidden layers do not fit beside two steps per layer: they take the variant without resident steps
  // (MRES_ = 0, TAIL & 64: every step streamed; up to 15 hidden layers -- architectures s02: 10 x 200).
  static constexpr int MRES = MRES_;
  static __host__ __device__ constexpr size_t res_off(int L) { return (small_bytes(L) + 1023) & ~(size_t)1023; }
  static __host__ __device__ constexpr size_t lds_bytes(int L) { return MRES > 0 ? res_off(L) + (size_t)L * 3 * MRES * 4 * 1024 : small_bytes(L); }
  static __host__ __device__ constexpr int max_layers() { int L = 1; while (L < 15 && lds_bytes(L + 1) <= 160 * 1024) ++L; return L; }
  static __host__ __device__ constexpr size_t section_off(int L) { return MlpTile4::section_off(L) + (size_t)L * MlpTile4::layer_floats(); }
  f32x4 ring[NT][4];
  float w0b, w0x, w0y;          // this lane's layer-0 row {b0, w00, w01}
  float *As, *Ns;
  const float *B1s, *wlS;
  const f32x4 *Wres;            // this wavefront's resident fragments in LDS: [layer][step < MRES][r][lane]
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, sec0, lbytes;
  int L, wave, lane, row, tpos;
#ifdef IONODE_STAMPS
  Stamps *sp;
#endif
  // fragment n = 4 s + r of this wavefront's stream of the layer at byte offset `lbase`
  __device__ __forceinline__ f32x4 frag(unsigned lbase, int n) const {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, lbase + (unsigned)n * 1024u, 0);
    return __builtin_bit_cast(f32x4, v);
  }
  // position of activation k in the transposed order of its k-tile: k = 16 kt + 4 q + r  ->  16 kt + 4 r + q
  static __device__ __forceinline__ int tp(int k) { return (k & ~15) + 4 * (k & 3) + ((k >> 2) & 3); }
  __device__ __forceinline__ void init(const KArgs &a, unsigned char *smem, int wave_, int lane_, int first_traj = 0) {
    L = a.L; wave = wave_; lane = lane_;
    const float *__restrict__ img = a.mlp + (a.traj_per_img > 0 ? (size_t)(first_traj / a.traj_per_img) * (size_t)a.mlp_stride : (size_t)0);
    row = (wave < 3) ? 64 * wave + lane : 192 + (lane & 15);
    tpos = tp(row);
    As = reinterpret_cast<float *>(smem);
    Ns = As + 2 * SLOTS * 16;
    float *b1 = Ns + 2 * NP;
    float *ws = b1 + (size_t)L * 256;
    const int tid = wave * 64 + lane;
    const size_t sec = section_off(L);
    for (int i = tid; i < L * 256; i += 256) b1[i] = img[sec + (size_t)(i >> 8) * layer_floats() + (size_t)(3 * FRAGS_FULL + FRAGS_REM) * 256 + (i & 255)];
    const float *wl = img + 4 * (size_t)NP + (size_t)L * ((size_t)4 * 43 * 256 + NP);   // behind the 16-column tile's layers (FRAGS = 43 per wavefront)
    for (int i = tid; i < NP + 4; i += 256) ws[i] = wl[i];
    B1s = b1; wlS = ws;
    w0b = img[4 * row + 0]; w0x = img[4 * row + 1]; w0y = img[4 * row + 2];
    const size_t img_bytes = (sec + (size_t)L * layer_floats()) * 4;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (int)img_bytes, 0x00020000);
    sec0 = (unsigned)(sec * 4);
    lbytes = (unsigned)(layer_floats() * 4);
    voff = (unsigned)(wave * FRAGS_FULL * 1024 + lane * 16);
    Wres = reinterpret_cast<const f32x4 *>(smem + res_off(L)) + (size_t)(wave < 3 ? wave : 0) * L * MRES * 4 * 64 + lane;
    if (MRES > 0 && wave < 3) {
      f32x4 *dst = reinterpret_cast<f32x4 *>(smem + res_off(L)) + (size_t)wave * L * MRES * 4 * 64 + lane;
      for (int l = 0; l < L; ++l) {
        f32x4 f[MRES > 0 ? MRES * 4 : 1];
#pragma unroll
        for (int n = 0; n < MRES * 4; ++n) f[n] = frag(sec0 + (unsigned)l * lbytes, n);
#pragma unroll
        for (int n = 0; n < MRES * 4; ++n) dst[((size_t)l * MRES * 4 + n) * 64] = f[n];
      }
    }
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) ring[s][r] = (L > 0 && (wave < 3 || s < 4)) ? frag(sec0, s * 4 + r) : f32x4{0, 0, 0, 0};
    __syncthreads();
  }
  // acc = fmaf(W[row][16 kt + 4 q + r], h[16 kt + 4 q + r], acc) for r = 0..3, q = 0..3 (the canonical order inside a k-tile): h[q] of lane r of the quad.
  // ONE asm statement per step: between two inline-asm statements hipcc inserts `s_nop 0`, which costs 4 cycles on top of the 9 of a dependent
  // v_fmac_f32 (tools/ubench/valu_chain.hip).  (A DPP source written by a VALU instruction needs two wait states: h comes straight from an LDS read.)
  static __device__ __forceinline__ void step16(float &acc, const f32x4 h, const f32x4 w0, const f32x4 w1, const f32x4 w2, const f32x4 w3) {
#define IONODE_R1_R(R, A, B, C, D) "v_fmac_f32_dpp %0, %1, " A " quad_perm:[" #R "," #R "," #R "," #R "] row_mask:0xf bank_mask:0xf\n\t" \
                                   "v_fmac_f32_dpp %0, %2, " B " quad_perm:[" #R "," #R "," #R "," #R "] row_mask:0xf bank_mask:0xf\n\t" \
                                   "v_fmac_f32_dpp %0, %3, " C " quad_perm:[" #R "," #R "," #R "," #R "] row_mask:0xf bank_mask:0xf\n\t" \
                                   "v_fmac_f32_dpp %0, %4, " D " quad_perm:[" #R "," #R "," #R "," #R "] row_mask:0xf bank_mask:0xf\n\t"
    asm(IONODE_R1_R(0, "%5", "%6", "%7", "%8") IONODE_R1_R(1, "%9", "%10", "%11", "%12") IONODE_R1_R(2, "%13", "%14", "%15", "%16") IONODE_R1_R(3, "%17", "%18", "%19", "%20")
        : "+v"(acc)
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]),
          "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]), "v"(w3[0]), "v"(w3[1]), "v"(w3[2]), "v"(w3[3]));
#undef IONODE_R1_R
  }
  // store an activation (row `row` at transposed position `tpos_`): transposed buffer (tiles 0..2 also at their second slot) and natural-order copy
  static __device__ __forceinline__ void put_h(float *__restrict__ A, float *__restrict__ N, int row_, int tpos_, float h) {
    A[tpos_] = h;
    if (row_ < 48) A[tpos_ + 16 * NT] = h;
    N[row_] = h;
  }
  // NS steps of this lane's chain: step s reads the lane's 16 bytes of slot (slot0 + s * STRIDE) and the ring's fragments 4 s .. 4 s + 3,
  // which are refilled for the coming layer right behind their last use
  // NRES: the first NRES steps' ring slots are refilled from the resident copy in LDS (`res`: the coming layer's fragments) instead of from L2
  template <int NS, int STRIDE, int NRES>
  __device__ __forceinline__ void walk(float &acc, const float *__restrict__ Hw, unsigned lnext, const f32x4 *__restrict__ res) {
    f32x4 hn = *reinterpret_cast<const f32x4 *>(Hw);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4 h = hn;
      if (s + 1 < NS) hn = *reinterpret_cast<const f32x4 *>(Hw + (s + 1) * STRIDE * 16);
      step16(acc, h, ring[s][0], ring[s][1], ring[s][2], ring[s][3]);
#ifndef IONODE_ROW1_NOREFILL   // timing experiment only (wrong results for L > 1): what the walk costs without its weight stream
#pragma unroll
      for (int r = 0; r < 4; ++r) ring[s][r] = (s < NRES) ? res[(s * 4 + r) * 64] : frag(lnext, s * 4 + r);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __device__ __forceinline__ float eval(float x0, float x1) {
    MSTAMP(0);
    {
      const float h = lrelu(fmaf(w0y, x1, fmaf(w0x, x0, w0b)));
      if (wave < 3 || lane < 16) put_h(As, Ns, row, tpos, h);
    }
    float acc_next = (L > 0) ? B1s[wave * 64 + lane] : 0.0f;
    __syncthreads();
    MSTAMP(1);
    const int c4 = (lane & 3) * 4, g16 = (lane >> 4) * 16;
    for (int l = 0; l < L; ++l) {
      const float *__restrict__ Ain = As + (l & 1) * SLOTS * 16;
      float *__restrict__ Aout = As + ((l + 1) & 1) * SLOTS * 16, *__restrict__ Nout = Ns + ((l + 1) & 1) * NP;
      const int ln = (l + 1 < L) ? l + 1 : 0;
      const unsigned lnext = sec0 + (unsigned)ln * lbytes;
      float acc = acc_next;
      if (l + 1 < L) acc_next = B1s[(l + 1) * 256 + wave * 64 + lane];
      MSTAMP(2);
      if (wave < 3) {
        // full rows: lane group g walks k-tile (s + g) mod 13 = slot s + g; the lane's 16 bytes of a slot: {h[16 kt + 4 q + (lane & 3)]}, q = 0..3
        walk<NT, 1, MRES>(acc, Ain + g16 + c4, lnext, Wres + (size_t)ln * MRES * 4 * 64);
        MSTAMP(3);
        put_h(Aout, Nout, row, tpos, lrelu(acc));
      } else {
        // remainder rows: lane group c runs partial chain c over the k-tiles c, c + 4, c + 8 (, 12: chain 0 only -- the others' step 3 reads
        // the duplicate slots 13..15 against -0.0f weights), then the four chains of a row meet across the lane groups: (p0 + p1) + (p2 + p3)
        walk<4, 4, 0>(acc, Ain + g16 + c4, lnext, Wres);
        MSTAMP(3);
        const float pair = acc + __shfl_xor(acc, 16);
        const float tot = pair + __shfl_xor(pair, 32);
        if (lane < 16) put_h(Aout, Nout, row, tpos, lrelu(tot));
      }
      __syncthreads();
      MSTAMP(4);
    }
    // Linear(N, 1): chain q = lane & 3 over k = 16 kt + 4 q + r, folded ((p0 + p1) + (p2 + p3)) + bl
    const float *__restrict__ Nin = Ns + (L & 1) * NP;
    f32x4 hl[NT], wv[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      hl[kt] = *reinterpret_cast<const f32x4 *>(Nin + 16 * kt + c4);
      wv[kt] = *reinterpret_cast<const f32x4 *>(wlS + 16 * kt + c4);
    }
    float part = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(wv[kt][r], hl[kt][r], part);
    }
    const float pair = part + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]: (p0 + p1) / (p2 + p3)
    const float out = (pair + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pair), 0x4E, 0xf, 0xf, false))) + wlS[NP];   // quad_perm [2,3,0,1]
    if ((L & 1) == 0) __syncthreads();   // the next evaluation's layer 0 rewrites buffer 0, which an even stack's output layer reads
    MSTAMP(5);
    return out;
  }
};

#ifndef IONODE_ROW1_MRES
#define IONODE_ROW1_MRES 2
#endif
using MlpRow1 = MlpRow1T<IONODE_ROW1_MRES>;   // nets of at most MlpRow1::max_layers() = 6 hidden layers
using MlpRow1Deep = MlpRow1T<0>;              // deeper stacks (TAIL & 64)

// ---------------------------------------------------------------------------------------------
// N = 10 nets (architectures s03-s05) at one trajectory per lane: the net evaluated PER LANE on the vector ALU, weights as
// SCALAR operands.  The MFMA form of this path (MlpTile::eval_tiny64) spends 80 MFMAs = 2560 cycles per evaluation on 16 x 16
// tiles of a 10 x 10 layer, gathers its inputs across lanes and keeps four accumulator tiles; per lane the net is 530 fmaf + 2 x 60
// LeakyReLU operations with no cross-lane traffic, and ~25 registers instead of ~110.  Every weight is used by all 64
// lanes at once, so it is read through the scalar cache (constant address space: s_load_dwordx8/x16) and enters the FMA as its
// one SGPR operand.  Same canonical order as the oracle / the MFMA tile with NT = 1:
//   hidden row j:  acc = bias; for r = 0..3: for q = 0..3: k = 4 q + r < N: acc = fmaf(W[j][k], h[k], acc)
//   Linear(N, 1):  part_q = 0; for r: k = 4 q + r < N: part_q = fmaf(wl[k], h[k], part_q); out = ((p0 + p1) + (p2 + p3)) + bl
// The padded terms the tile executes (k >= N: fmaf(0, 0, acc)) are skipped: they return acc for every acc except -0, and an
// accumulator can only be -0 if its bias is -0 (x + (-x) rounds to +0; +0 + -0 = +0), which ionode_mlp_pack rules out by writing
// bias + 0.0f into this section (N < 16; the tile's own trailing padded term does the same to its result).
// TWO ROWS PER INSTRUCTION: rows 2 m and 2 m + 1 run the same k sequence on the same inputs, so their chains are the two halves of
// one v_pk_fma_f32 -- weights {W[2m][k], W[2m+1][k]} in an SGPR pair, h[k] broadcast from its half of the activation pair
// (op_sel), accumulators in a VGPR pair: one exact fmaf per half, 4 cycles for both (gfx950's vector fp32 peak IS the packed rate).
// A layer's output pair m = {h[2m], h[2m+1]} is the next layer's input pair.  The LeakyReLU multiply is packed as well.
// Image section (ionode_mlp_pack, behind wl / bl): row pair m of layer 0: {b0, b0'} {w00, w00'} {w01, w01'} {0, 0}; then, per hidden
// layer and row pair, PB floats: {W[2m][k], W[2m+1][k]} in the canonical k order, {bias, bias'}, pad.
// ---------------------------------------------------------------------------------------------
#ifndef IONODE_VNET_PAIRS
#define IONODE_VNET_PAIRS 5   // row pairs of a hidden layer evaluated together (scalar loads of the group in flight at once, independent chains): 65 536 x 20 001: 15.5 ms at 1, 14.3 at 2, 14.2 at 3, 13.6 at 5; 262 144: 38.4 / 36.6 / 36.6 / 36.0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// acc + w * h.lo / acc + w * h.hi in both halves (one fused multiply-add each); w: SGPR pair
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 w, f32x2 h, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(h));
  return acc;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 w, f32x2 h, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(h));
  return acc;
}
__device__ __forceinline__ f32x2 lrelu2(f32x2 x) {
  f32x2 t, h;
  const f32x2 c = {0.01f, 0.01f};
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "s"(c));
  float h0, h1;
  asm("v_max_f32 %0, %1, %2" : "=v"(h0) : "v"(x.x), "v"(t.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(h1) : "v"(x.y), "v"(t.y));
  h.x = h0; h.y = h1;
  return h;
}
template <int N> struct MlpLane {
  static_assert(N == 10, "the per-lane net is instantiated for N = 10 (architectures s03-s05)");
  static constexpr int GW = 1;
  static constexpr int NP = 16;
  static constexpr int NPAIR = (N + 1) / 2;
  static constexpr int PB = (2 * (N + 1) + 3) & ~3;   // floats per (layer, row pair) block of the scalar section
  typedef const float __attribute__((address_space(4))) cfloat;   // constant address space: uniform loads are scalar loads
  typedef const f32x2 __attribute__((address_space(4))) cfloat2;
  const cfloat *img;   // the tile's packed image
  int L;
#ifdef IONODE_STAMPS
  Stamps *sp;
#endif
  static __host__ __device__ constexpr size_t lds_bytes(int) { return 0; }
  static __host__ __device__ constexpr size_t scalar_floats(int L) { return (size_t)NPAIR * 8 + (size_t)L * NPAIR * PB; }
  __device__ __forceinline__ void init(const KArgs &a, unsigned char *, int, int, int first_traj = 0) {
    L = a.L;
    const float *g = a.mlp + (a.traj_per_img > 0 ? (size_t)(first_traj / a.traj_per_img) * (size_t)a.mlp_stride : (size_t)0);
    img = (const cfloat *)(uintptr_t)g;
  }
  // canonical position of k in a row's chain: r-major, q-minor over k = 4 q + r < N
  static __host__ __device__ constexpr int k_at(int pos) {
    int n = 0;
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q)
        if (4 * q + r < N) { if (n == pos) return 4 * q + r; ++n; }
    return -1;
  }
#ifndef IONODE_VNET_RELOAD
#define IONODE_VNET_RELOAD 1   // 1: Linear(2, N) and Linear(N, 1) are scalar loads of THIS evaluation (round 5).  0 (rounds 3-4): hipcc hoists
                               // the 62 loop-invariant scalars out of the attempt loop, cannot keep them in scalar registers next to the hidden
                               // layers' 110 and parks them in VGPR lanes: 79 v_readlane per evaluation -- vector-ALU work in a vector-issue-bound kernel
#endif
  __device__ __forceinline__ float eval_tiny64(float x0, float x1) {
    constexpr size_t lstride = (size_t)256 + NP;  // MlpTile<1, 1, 1, 1>::layer_floats(): one fragment + bias[NP]
    const cfloat *im = img;
    if (IONODE_VNET_RELOAD) asm volatile("" : "+s"(im));     // (an opaque copy of the pointer: loads through it stay inside this evaluation)
    const cfloat *wl = im + 4 * NP + (size_t)L * lstride;   // wl[NP], bl, 3 pad
    const cfloat2 *s0 = reinterpret_cast<const cfloat2 *>(wl + NP + 4);   // the scalar section: layer 0 ...
    const cfloat2 *sh = s0 + NPAIR * 4;                                    // ... and the hidden layers
    f32x2 h[NPAIR];
    {
      const f32x2 xx = {x0, x1};
#pragma unroll
      for (int m = 0; m < NPAIR; ++m) h[m] = lrelu2(pk_fma_hi(s0[4 * m + 2], xx, pk_fma_lo(s0[4 * m + 1], xx, s0[4 * m + 0])));
    }
    constexpr int GP = IONODE_VNET_PAIRS;
    for (int l = 0; l < L; ++l) {
      f32x2 g[NPAIR];
#pragma unroll
      for (int m0 = 0; m0 < NPAIR; m0 += GP) {
        // a group's scalar loads are issued together (one wait); its chains are independent of each other (a lone dependent
        // chain stalls a SIMD that holds few wavefronts)
#pragma unroll
        for (int u = 0; u < GP; ++u)
          if (m0 + u < NPAIR) g[m0 + u] = sh[((size_t)l * NPAIR + m0 + u) * (PB / 2) + N];
#ifndef IONODE_VNET_SPLIT
#define IONODE_VNET_SPLIT 5   // > 0: a scheduling barrier after this many k positions: half of the 110 weight scalars of a row-pair group in flight, so that the kernel's own uniform state stays in scalar registers (with IONODE_VNET_RELOAD: 539 -> 49 v_readlane per attempt)
#endif
#pragma unroll
        for (int pos = 0; pos < N; ++pos) {
#pragma unroll
          for (int u = 0; u < GP; ++u)
            if (m0 + u < NPAIR) {
              const f32x2 w = sh[((size_t)l * NPAIR + m0 + u) * (PB / 2) + pos];
              const int k = k_at(pos);
              g[m0 + u] = (k & 1) ? pk_fma_hi(w, h[k >> 1], g[m0 + u]) : pk_fma_lo(w, h[k >> 1], g[m0 + u]);
            }
          if (IONODE_VNET_SPLIT > 0 && pos + 1 == IONODE_VNET_SPLIT) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < GP; ++u)
          if (m0 + u < NPAIR) g[m0 + u] = lrelu2(g[m0 + u]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < NPAIR; ++m) h[m] = g[m];
    }
    if (IONODE_VNET_RELOAD) asm volatile("" : "+s"(wl));     // (Linear(N, 1)'s scalars are loaded after the hidden stack, not carried through it)
    float part[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      part[q] = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < N) part[q] = fmaf(wl[4 * q + r], h[(4 * q + r) >> 1][(4 * q + r) & 1], part[q]);
    }
    return ((part[0] + part[1]) + (part[2] + part[3])) + wl[NP];
  }
  __device__ __forceinline__ float eval(float x0, float x1) { return eval_tiny64(x0, x1); }
};

// ---------------------------------------------------------------------------------------------
// ANY width up to 512 (round 5): the 16-trajectory MFMA tile with the k-tile count NT = ceil(N / 16) as a RUN-TIME value.  The widths of
// architectures/s00-s11.py (N = 10, 100, 200, 500) have their own tuned tiles above; table-s1.py:145-153 builds Linear(2, N) ... Linear(N, 1)
// for any (n_layers, n_nodes), and a user's --info file with N = 50 or 64 used to fall out of the fused path with an error.  No performance
// target: weights are read from L2 as they are needed (one 16-byte load per lane and k-tile, the next one in flight), no register ring, no
// generated stream.  SAME canonical accumulation order as every other form ("canonical arithmetic"): four wavefronts; NT = 4 F + R;
// wavefront w owns the full row tiles rt = w, w + 4, ... < 4 F -- ONE chain seeded with the bias over the k-tiles in the rotated order
// kt = (s + w) mod NT -- and partial chain w (k-tiles kt % 4 == w, ascending; chain 0 carries the bias) of each of the R remainder row
// tiles, folded (p0 + p1) + (p2 + p3); Linear(N, 1): four chains by q.  lane = 16 q + m: A fragment = W[16 rt + m][16 kt + 4 q + r] over r,
// B operand / accumulator = h[16 kt + 4 q + r][trajectory m].
// Image (ionode_mlp_pack for widths without a tuned tile): [NP][4]{b0, w00, w01, 0} | L x ([rt][kt][lane] float4 over r, then bias[NP]) |
// wl[NP], bl, 3 pad.
// ---------------------------------------------------------------------------------------------
struct MlpGen {
  static constexpr int GW = 4;
  static constexpr int NT_MAX = 32;
  f32x4 *Hs, *Ps;
  const f32x4 *W0s;
  const float *biasS, *wlS;
  const f32x4 *hid;    // hidden layer 0 in the image (global memory / L2)
  size_t lstride4;     // float4 per hidden layer in the image
  int L, NT, NP, F4, R, wave, lane;
#ifdef IONODE_STAMPS
  Stamps *sp;
#endif
  static __host__ __device__ constexpr size_t layer_floats(int NT) { return (size_t)NT * NT * 256 + (size_t)16 * NT; }
  static __host__ __device__ constexpr size_t image_floats(int L, int NT) { return (size_t)4 * 16 * NT + (size_t)L * layer_floats(NT) + (size_t)16 * NT + 4; }
  // activations x2, partial sums of the (at most three) remainder tiles x2, layer-0 rows; biases, output weights
  static __host__ __device__ constexpr size_t lds_bytes(int L, int NT) {
    return ((size_t)2 * NT * 64 + (size_t)2 * 3 * 4 * 64 + (size_t)16 * NT) * 16 + ((size_t)L * 16 * NT + (size_t)16 * NT + 4) * 4;
  }
  __device__ __forceinline__ void init(const KArgs &a, unsigned char *smem, int wave_, int lane_, int first_traj = 0) {
    L = a.L; NT = a.NT; NP = 16 * NT; wave = wave_; lane = lane_;
    F4 = 4 * (NT / 4); R = NT - F4;
    const float *__restrict__ img = a.mlp + (a.traj_per_img > 0 ? (size_t)(first_traj / a.traj_per_img) * (size_t)a.mlp_stride : (size_t)0);
    Hs = reinterpret_cast<f32x4 *>(smem);
    Ps = Hs + 2 * NT * 64;
    f32x4 *w0 = Ps + 2 * 3 * 4 * 64;
    float *bs = reinterpret_cast<float *>(w0 + NP);
    float *ws = bs + (size_t)L * NP;
    const size_t lstride = layer_floats(NT);
    const int tid = wave * 64 + lane;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(img);
    for (int i = tid; i < NP; i += 256) w0[i] = src[i];
    for (int i = tid; i < L * NP; i += 256) bs[i] = img[4 * (size_t)NP + (size_t)(i / NP) * lstride + (lstride - NP) + (i % NP)];
    const float *wl = img + 4 * (size_t)NP + (size_t)L * lstride;
    for (int i = tid; i < NP + 4; i += 256) ws[i] = wl[i];
    W0s = w0; biasS = bs; wlS = ws;
    hid = src + NP;
    lstride4 = lstride / 4;
    __syncthreads();
  }
  __device__ __forceinline__ float eval(float x0, float x1) {
    const int q = lane >> 4;
    MSTAMP(0);
    // layer 0: Linear(2, N) + LeakyReLU, row tile rt by wavefront rt % 4, accumulator layout
    for (int rt = wave; rt < NT; rt += 4) {
      f32x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 w = W0s[16 * rt + 4 * q + r];
        h[r] = lrelu(fmaf(w[2], x1, fmaf(w[1], x0, w[0])));
      }
      Hs[rt * 64 + lane] = h;
    }
    __syncthreads();
    MSTAMP(1);
    for (int l = 0; l < L; ++l) {
      const f32x4 *__restrict__ Hin = Hs + (l & 1) * NT * 64 + lane;
      f32x4 *__restrict__ Hout = Hs + ((l + 1) & 1) * NT * 64 + lane;
      f32x4 *__restrict__ Pl = Ps + (l & 1) * 3 * 4 * 64 + lane;
      const f32x4 *__restrict__ Wl = hid + (size_t)l * lstride4 + lane;
      const float *__restrict__ bl_ = biasS + l * NP + 4 * q;
      // full row tiles: one chain each, k-tiles in the rotated order (s + wave) mod NT; the next fragment and B operand are in flight
      for (int rt = wave; rt < F4; rt += 4) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(bl_ + 16 * rt);
        const f32x4 *__restrict__ Wr = Wl + (size_t)rt * NT * 64;
        int kt = wave;   // (wave < 4 <= F4 <= NT)
        f32x4 a_n = Wr[kt * 64], b_n = Hin[kt * 64];
        for (int s = 0; s < NT; ++s) {
          const f32x4 av = a_n, bv = b_n;
          kt = (kt + 1 == NT) ? 0 : kt + 1;
          if (s + 1 < NT) { a_n = Wr[kt * 64]; b_n = Hin[kt * 64]; }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0);
        }
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu(acc[r]);
        Hout[rt * 64] = h;
      }
      // remainder row tiles: partial chain `wave` over the k-tiles kt % 4 == wave, ascending (chain 0 is seeded with the bias)
      for (int j = 0; j < R; ++j) {
        const int rt = F4 + j;
        f32x4 acc = f32x4{0, 0, 0, 0};
        if (wave == 0) acc = *reinterpret_cast<const f32x4 *>(bl_ + 16 * rt);
        const f32x4 *__restrict__ Wr = Wl + (size_t)rt * NT * 64;
        for (int kt = wave; kt < NT; kt += 4) {
          const f32x4 av = Wr[kt * 64], bv = Hin[kt * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc, 0, 0, 0);
        }
        Pl[(j * 4 + wave) * 64] = acc;
      }
      __syncthreads();
      // every wavefront folds the remainder tiles itself (identical bits; each reads the slot after its own write)
      for (int j = 0; j < R; ++j) {
        const f32x4 p0 = Pl[(j * 4 + 0) * 64], p1 = Pl[(j * 4 + 1) * 64], p2 = Pl[(j * 4 + 2) * 64], p3 = Pl[(j * 4 + 3) * 64];
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = lrelu((p0[r] + p1[r]) + (p2[r] + p3[r]));
        Hout[(F4 + j) * 64] = h;
      }
      MSTAMP(3);
    }
    // Linear(N, 1): four partial chains (one per lane group q), fixed combine tree
    const f32x4 *__restrict__ Hin = Hs + (L & 1) * NT * 64 + lane;
    float part = 0.0f;
    for (int kt = 0; kt < NT; ++kt) {
      const f32x4 w = *reinterpret_cast<const f32x4 *>(wlS + 16 * kt + 4 * q);
      const f32x4 h = Hin[kt * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(w[r], h[r], part);
    }
    const float pair = part + __shfl_xor(part, 16);
    const float out = (pair + __shfl_xor(pair, 32)) + wlS[NP];
    if ((L & 1) == 0) __syncthreads();   // the next evaluation's layer 0 rewrites buffer 0
    MSTAMP(5);
    return out;
  }
};

// Closed-form models carry an empty stand-in so the integrator code is shared.
struct NoMlp {
  static constexpr int GW = 1;
  __device__ __forceinline__ float eval(float, float) { return 0.0f; }
};

// ---------------------------------------------------------------------------------------------
// func.forward(t, y) of the reference, per lane.  The protocol voltage at the stage time (and whether
// the time was inside the protocol's range) is looked up by the caller, ahead of the stage.
// ---------------------------------------------------------------------------------------------
// (rate constants of one stage voltage; see closed_rates() below)
template <int MODEL> struct ClosedRates {
  static constexpr int NR = (MODEL == IONODE_MODEL_MARKOV6) ? 6 : 4;
  double k[NR];
  float kf[NR];
  bool oob32;
};
template <int MODEL, typename S, bool WIDE = false, typename MLP>
__device__ __forceinline__ void rhs(const KArgs &a, const double *p, double v, bool inrange, const S *y, S *f,
                                    MLP &mlp, ClosedRates<IONODE_MODEL_HH2> *cr = nullptr, bool fresh = true) {
  using MT = ModelTraits<MODEL>;
  constexpr bool F32 = sizeof(S) == 4;

  if constexpr (MODEL == IONODE_MODEL_MARKOV6) {
    if (F32 && !inrange) {
      // v = torch.tensor([-80]) is int64: `p * v` is float32 and exp runs in fp32 (train-d1.py:169-178)
      const float vf = (float)a.v_oob;
      const float a1 = (float)p[0] * det_expf((float)p[1] * vf);
      const float b1 = (float)p[2] * det_expf((float)(-p[3]) * vf);
      const float bh = (float)p[4] * det_expf((float)p[5] * vf);
      const float ah = (float)p[6] * det_expf((float)(-p[7]) * vf);
      const float a2 = (float)p[8] * det_expf((float)p[9] * vf);
      const float b2 = (float)p[10] * det_expf((float)(-p[11]) * vf);
      const float c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
      f[0] = a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1;
      f[1] = b1 * c1 + ah * ic2 - (a1 + bh) * c2;
      f[2] = a2 * ic1 + bh * o - (b2 + ah) * i_;
      f[3] = a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1;
      f[4] = b1 * ic1 + bh * c2 - (ah + a1) * ic2;
      f[5] = a2 * c1 + ah * i_ - (b2 + bh) * o;
      return;
    }
    const double a1 = p[0] * det_exp(p[1] * v);
    const double b1 = p[2] * det_exp(-p[3] * v);
    const double bh = p[4] * det_exp(p[5] * v);
    const double ah = p[6] * det_exp(-p[7] * v);
    const double a2 = p[8] * det_exp(p[9] * v);
    const double b2 = p[10] * det_exp(-p[11] * v);
    const double c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
    f[0] = (S)(a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1);
    f[1] = (S)(b1 * c1 + ah * ic2 - (a1 + bh) * c2);
    f[2] = (S)(a2 * ic1 + bh * o - (b2 + ah) * i_);
    f[3] = (S)(a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1);
    f[4] = (S)(b1 * ic1 + bh * c2 - (ah + a1) * ic2);
    f[5] = (S)(a2 * c1 + ah * i_ - (b2 + bh) * o);
    return;
  } else {
    constexpr bool HAS_HH_A = (MODEL == IONODE_MODEL_HH2 || MODEL == IONODE_MODEL_NND);
    const S av = y[0], rv = y[1];
    const bool oob32 = F32 && !inrange;

    // MLP term first: it is a tile-wide collective, so every lane takes part whatever its branch below
    float net = 0.0f;
    if constexpr (MT::MLP) {
      const float vf = (float)a.v_oob;
      // v / self.vrange, then .float(); net / self.netscale -- exact quotients by the constants 100 and 1000 (div_const)
      const float nv = oob32 ? vf / 100.0f : (float)div_const(v, 100.0, 0.01);
      if constexpr (WIDE) net = div_constf(mlp.eval_tiny64(nv, (float)av), 1000.0f, 0.001f);  // 64 trajectories per wavefront (N <= 16)
      else net = div_constf(mlp.eval(nv, (float)av), 1000.0f, 0.001f);
    }

    if (oob32) {
      const float vf = (float)a.v_oob;
      const float af = (float)av, rf = (float)rv;
      const float k3 = (float)p[4] * det_expf((float)p[5] * vf);
      const float k4 = (float)p[6] * det_expf((float)(-p[7]) * vf);
      const float drdt = -k3 * rf + k4 * (1.0f - rf);
      float dadt = 0.0f;
      if constexpr (HAS_HH_A) {
        const float k1 = (float)p[0] * det_expf((float)p[1] * vf);
        const float k2 = (float)p[2] * det_expf((float)(-p[3]) * vf);
        dadt = k1 * (1.0f - af) - k2 * af;
      }
      if constexpr (MT::MLP) dadt = (MODEL == IONODE_MODEL_NND) ? dadt + net : net;
      f[0] = (S)dadt;
      f[1] = (S)drdt;
      return;
    }
    const S one_m_a = (S)1 - av;  // `1. - a` / `self.unity - r` are formed in y.dtype
    const S one_m_r = (S)1 - rv;
    // (not for the 64-per-wavefront N <= 16 kernel: two interleaved branch-free exps cost ~30 registers -- it went from 252 to 284
    // VGPRs, i.e. from two wavefronts per SIMD to one, 58 -> 87 ms)
    // (the 4-trajectory tile keeps a 208-register weight ring: the branchy form with one exp in flight, same bits)
    constexpr bool TIGHT = std::is_same<MLP, MlpTile4>::value || std::is_same<MLP, MlpRow1>::value || std::is_same<MLP, MlpRow1Deep>::value;
    auto dexp = [](double x) { if constexpr (TIGHT) return det_exp_ldexp(x); else if constexpr (MT::MLP && !WIDE) return det_exp_s(x); else return det_exp(x); };
    double k3, k4, dadt = 0.0;
    if constexpr (MT::MLP && WIDE) {
      // one trajectory per lane (N <= 16): the closed-form kernels' exp -- addend constants as scalar operands, one v_ldexp_f64, and
      // the three range cases skipped when every lane's arguments are in range (closed_rates); same operations, same bits
      // (round 5) the rates depend on the stage VOLTAGE only: the integrator says `fresh = false` when every lane of the wavefront sees the
      // previous stage's voltage again (stage 6 always; every stage on a protocol's plateaus) and the products kept in *cr are reused
      constexpr int NX = HAS_HH_A ? 4 : 2;
      double kk[4];
      if (fresh || cr == nullptr) {
        double x[NX], e[NX];
        x[0] = p[5] * v; x[1] = -p[7] * v;
        if constexpr (HAS_HH_A) { x[2] = p[1] * v; x[3] = -p[3] * v; }
        bool in = true;
#pragma unroll
        for (int i = 0; i < NX; ++i) in = in && (__builtin_fabs(x[i]) <= 708.0);
        if (__ballot(!in) == 0ull) {
#pragma unroll
          for (int i = 0; i < NX; ++i) e[i] = det_exp_inrange(x[i]);
        } else {
#pragma unroll
          for (int i = 0; i < NX; ++i) e[i] = det_exp_ldexp(x[i]);
        }
        kk[2] = p[4] * e[0]; kk[3] = p[6] * e[1];
        if constexpr (HAS_HH_A) { kk[0] = p[0] * e[2]; kk[1] = p[2] * e[3]; }
        if (cr != nullptr) {
          cr->k[2] = kk[2]; cr->k[3] = kk[3];
          if constexpr (HAS_HH_A) { cr->k[0] = kk[0]; cr->k[1] = kk[1]; }
        }
      } else {
        kk[2] = cr->k[2]; kk[3] = cr->k[3];
        if constexpr (HAS_HH_A) { kk[0] = cr->k[0]; kk[1] = cr->k[1]; }
      }
      k3 = kk[2]; k4 = kk[3];
      if constexpr (HAS_HH_A) dadt = kk[0] * (double)one_m_a - kk[1] * (double)av;
    } else {
      k3 = p[4] * dexp(p[5] * v);
      k4 = p[6] * dexp(-p[7] * v);
      if constexpr (HAS_HH_A) {
        const double k1 = p[0] * dexp(p[1] * v);
        const double k2 = p[2] * dexp(-p[3] * v);
        dadt = k1 * (double)one_m_a - k2 * (double)av;
      }
    }
    const double drdt = -k3 * (double)rv + k4 * (double)one_m_r;
    if constexpr (MT::MLP) dadt = (MODEL == IONODE_MODEL_NND) ? dadt + (double)net : (double)net;
    f[0] = (S)dadt;
    f[1] = (S)drdt;
  }
}

// Closed-form models, split form of rhs(): the rate constants depend on the stage VOLTAGE only, and the last two stages of a
// dopri5 attempt share their time (alpha = 1, 1), so the integrator evaluates them once for both (4 of 24 exp per attempt for
// the 2-state model, 12 of 72 for the 6-state model).  Same expressions as rhs(), same bits.
template <int MODEL, typename S>
__device__ __forceinline__ void closed_rates(const KArgs &a, const double *p, double v, bool inrange, ClosedRates<MODEL> &R) {
  constexpr int NR = ClosedRates<MODEL>::NR;
  R.oob32 = (sizeof(S) == 4) && !inrange;
  if (R.oob32) {
    const float vf = (float)a.v_oob;  // int64 tensor([-80]): `p * v` is float32 and exp runs in fp32
#pragma unroll
    for (int i = 0; i < NR; ++i) R.kf[i] = (float)p[2 * i] * det_expf((float)((i & 1) ? -p[2 * i + 1] : p[2 * i + 1]) * vf);
  } else {
    double x[NR];
    bool inr = true;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      x[i] = ((i & 1) ? -p[2 * i + 1] : p[2 * i + 1]) * v;
      inr = inr && (__builtin_fabs(x[i]) <= 708.0);
    }
    if (__ballot(!inr) == 0ull) {   // every argument of every lane in range: none of exp's special cases can apply (wave-uniform branch)
#pragma unroll
      for (int i = 0; i < NR; ++i) R.k[i] = p[2 * i] * det_exp_inrange(x[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) R.k[i] = p[2 * i] * det_exp_ldexp(x[i]);
    }
  }
}
template <int MODEL, typename S>
__device__ __forceinline__ void closed_rhs(const ClosedRates<MODEL> &R, const S *y, S *f) {
  if constexpr (MODEL == IONODE_MODEL_MARKOV6) {
    if (R.oob32) {
      const float a1 = R.kf[0], b1 = R.kf[1], bh = R.kf[2], ah = R.kf[3], a2 = R.kf[4], b2 = R.kf[5];
      const float c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
      f[0] = a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1;
      f[1] = b1 * c1 + ah * ic2 - (a1 + bh) * c2;
      f[2] = a2 * ic1 + bh * o - (b2 + ah) * i_;
      f[3] = a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1;
      f[4] = b1 * ic1 + bh * c2 - (ah + a1) * ic2;
      f[5] = a2 * c1 + ah * i_ - (b2 + bh) * o;
      return;
    }
    const double a1 = R.k[0], b1 = R.k[1], bh = R.k[2], ah = R.k[3], a2 = R.k[4], b2 = R.k[5];
    const double c1 = y[0], c2 = y[1], i_ = y[2], ic1 = y[3], ic2 = y[4], o = y[5];
    f[0] = (S)(a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1);
    f[1] = (S)(b1 * c1 + ah * ic2 - (a1 + bh) * c2);
    f[2] = (S)(a2 * ic1 + bh * o - (b2 + ah) * i_);
    f[3] = (S)(a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1);
    f[4] = (S)(b1 * ic1 + bh * c2 - (ah + a1) * ic2);
    f[5] = (S)(a2 * c1 + ah * i_ - (b2 + bh) * o);
  } else {
    const S av = y[0], rv = y[1];
    if (R.oob32) {
      const float af = (float)av, rf = (float)rv;
      const float drdt = -R.kf[2] * rf + R.kf[3] * (1.0f - rf);
      const float dadt = R.kf[0] * (1.0f - af) - R.kf[1] * af;
      f[0] = (S)dadt;
      f[1] = (S)drdt;
      return;
    }
    const S one_m_a = (S)1 - av, one_m_r = (S)1 - rv;
    const double drdt = -R.k[2] * (double)rv + R.k[3] * (double)one_m_r;
    const double dadt = R.k[0] * (double)one_m_a - R.k[1] * (double)av;
    f[0] = (S)dadt;
    f[1] = (S)drdt;
  }
}

template <typename S, int D> __device__ __forceinline__ S rms_norm(const S *x) {
  S s = x[0] * x[0];
#pragma unroll
  for (int i = 1; i < D; ++i) s = s + x[i] * x[i];
  s = s / (S)D;
  return Real<S>::sqrt_(s);
}
template <typename S> __device__ __forceinline__ S abs_(S x) { return x < 0 ? -x : x; }

// Wavefronts per SIMD asked of hipcc (__launch_bounds__).  2-state closed-form kernels: TWO -- a 256-register budget, of which hipcc
// uses 118 (lean variant: FOUR resident per SIMD), 125-131 (table variant) or 150-158 (general: three per SIMD) -- round 4: constants
// as scalar operands, lane- and parameter-derived invariants kept out of the attempt loop, plain work-list emission; asked for three,
// hipcc's scheduler fills the 168 and spills 2-6 registers to scratch.  6-state: ONE (the whole register file; the lean variant
// comes out at 232 -- two resident per SIMD): asked for two, the general variant spilled 48 dwords into scratch inside the stage loop
// and ran 1.6x (65 536 trajectories) to 2x (16 384) slower.  MLP tiles: 1 per SIMD.
#ifndef IONODE_M6_WAVES
#define IONODE_M6_WAVES 1
#endif
#ifndef IONODE_CF_WAVES
#ifndef IONODE_HH2_WAVES
#define IONODE_HH2_WAVES 2
#endif
#define IONODE_CF_WAVES(MODEL, G) ((G) > 1 ? 1 : ((MODEL) == IONODE_MODEL_HH2 ? IONODE_HH2_WAVES : IONODE_M6_WAVES))
#endif
#ifndef IONODE_T64_WAVES
#define IONODE_T64_WAVES 1
#endif
#ifndef IONODE_N100_WAVES
#define IONODE_N100_WAVES 1   // N = 100 tile (NT = 7): 2 = ask hipcc for a 256-register build, two tiles per compute unit (A/B)
#endif
#define IONODE_WAVES_PER_SIMD(MODEL, G, NT, RT) \
  (((MODEL) >= IONODE_MODEL_NNF && (RT) == 64) ? IONODE_T64_WAVES : (((MODEL) >= IONODE_MODEL_NNF && (NT) == 7) ? IONODE_N100_WAVES : IONODE_CF_WAVES(MODEL, G)))

template <typename S, int D> __device__ __forceinline__ void store_state(S *dst, const S *v) {
  if constexpr (D == 2 && sizeof(S) == 8) {
    *reinterpret_cast<double2 *>(dst) = make_double2(v[0], v[1]);
  } else if constexpr (D == 2 && sizeof(S) == 4) {
    *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
  } else if constexpr (sizeof(S) == 8) {
#pragma unroll
    for (int d = 0; d < D; d += 2) *reinterpret_cast<double2 *>(dst + d) = make_double2(v[d], v[d + 1]);
  } else {
#pragma unroll
    for (int d = 0; d < D; d += 2) *reinterpret_cast<float2 *>(dst + d) = make_float2(v[d], v[d + 1]);
  }
}

// ---------------------------------------------------------------------------------------------
// The integrator.  One workgroup = one tile of TPW trajectories (G wavefronts for MLP models).
// ---------------------------------------------------------------------------------------------
// Lane-wise kernels (closed-form models, N <= 16 nets at 64 per wavefront): a tile is ONE wavefront, but a workgroup carries FOUR independent
// tiles (IONODE_LW_TILES_PER_WG), one per SIMD of a compute unit, each with its own LDS region and no barrier between them.  The
// hardware hands out whole workgroups, so the plan can cap the wavefronts per SIMD of a small launch through the workgroup's LDS
// reservation (ionode_capi.hip even_placement): single-wavefront workgroups of a launch that does not fill the chip get stacked three
// deep on some SIMDs while others idle (6-state, 65 536 trajectories: 24.2 ms stacked, 20.0 ms at one per SIMD).
#define IONODE_LW_TILES_PER_WG 4
#define IONODE_IS_LW(MODEL, RT) ((MODEL) == IONODE_MODEL_HH2 || (MODEL) == IONODE_MODEL_MARKOV6 || (RT) == 64)
template <int MODEL, typename S, int G, int RT, int NT, int PD, int TAIL>
__global__ void __launch_bounds__(64 * (IONODE_IS_LW(MODEL, RT) ? IONODE_LW_TILES_PER_WG : G), IONODE_WAVES_PER_SIMD(MODEL, G, NT, RT)) ionode_dopri5_kernel(const KArgs a_in) {
  using MT = ModelTraits<MODEL>;
  // Per-variant CONTRACTS (ionode_capi.hip make_plan selects a variant only when they hold).  What a variant is never asked to do is
  // cleared in its private copy of the arguments: the branches fold away at compile time, and with them their code, their registers
  // and the scalars (pointers, caps) that would otherwise stay live through the attempt loop -- in SGPRs that spill into VGPR lanes.
  //   TAIL == 1 of a lane-wise kernel (LEAN): uniform protocol grid, VERIFIED uniform output grid, states only (no current trace, no
  //                fused objective), no step log, no checkpoints
  //   TAIL == 2 of a closed-form kernel (table variant): uniform protocol grid, no step log, no checkpoints
  constexpr bool LEAN = IONODE_LEAN && (!MT::MLP || RT == 64) && TAIL == 1;
  constexpr bool LEANT = IONODE_LEAN && !MT::MLP && TAIL == 2;
  //   TAIL & 8 of an MLP tile kernel: uniform protocol grid, VERIFIED uniform output grid, no step log, no checkpoints (states,
  //                current trace and fused objective stay run-time choices)
  constexpr bool LEANM = IONODE_LEAN && MT::MLP && G > 1 && (TAIL & 8);
  KArgs a = a_in;
  if constexpr (LEANM) a.te_exact = 1;
  if constexpr (LEAN || LEANT || LEANM) { a.prot_t = nullptr; a.step_log = nullptr; a.step_log_cap = 0; a.ckpt = nullptr; a.ckpt_cap = 0; }
  if constexpr (LEAN) { a.i_out = nullptr; a.sse_out = nullptr; a.sse_ref = nullptr; a.v_tab = nullptr; a.te_exact = 1; }
#ifdef IONODE_STAMPS
  a.step_log = a_in.step_log; a.step_log_cap = a_in.step_log_cap;   // (the diagnostic build reports its stamps through the step log)
#endif
  using R = Real<S>;
  constexpr int D = MT::D, NPAR = MT::NPAR;
  // trajectories per wavefront: 16 for MLP tiles (MFMA column count); closed-form kernels: 64 (one per lane), or RT
  // (16: lanes replicated 4x) for small batches, where 4x more wavefronts matter more than lane efficiency
  // N <= 16 nets also at 64 per wavefront: RT slot == 64 of an MLP kernel (MlpTile::eval_tiny64).  Such a kernel is
  // "lane-wise" (LW) like the closed-form ones -- one trajectory per lane -- and shares their dense-output machinery:
  // interpolant rows in LDS (behind the MlpTile region), arithmetic output times, carried stage voltages, work-list
  // emission.
  constexpr bool T64 = MT::MLP && RT == 64;
  static_assert(!T64 || (G == 1 && NT == 1), "64 trajectories per wavefront is the resident-weights (N <= 16) path");
  constexpr bool LW = !MT::MLP || T64;
  // MLP tile kernels with TAIL == 4: two 16-trajectory column sets per workgroup (MlpTile::NSETS); wavefronts [0, WPS) integrate
  // set 0, [WPS, 2 WPS) set 1.  LPS = lanes of a wavefront that hold distinct trajectories.
  constexpr int NSETS = (MT::MLP && G > 1 && (TAIL & 4)) ? 2 : 1;   // (TAIL & 8: the tile kernels' lean variant, top of the kernel)
  constexpr int WPS = G / NSETS;
  // TAIL & 16 of an N = 200 tile kernel: FOUR trajectories per tile (MlpTile4: small batches and single calls); lane = 4 b + j holds trajectory j
  constexpr bool T4 = MT::MLP && G == 4 && NT == 13 && (TAIL & 16);
  static_assert(!T4 || NSETS == 1, "the 4-trajectory tile has one column set");
  // TAIL & 32 of an N = 200 tile kernel: ONE trajectory per tile (MlpRow1: a lane owns a row; the reference's own call shape); every lane holds the trajectory
  constexpr bool T1 = MT::MLP && G == 4 && NT == 13 && (TAIL & 32);
  static_assert(!T1 || (NSETS == 1 && !T4), "the one-trajectory tile has one column set and is not the 4-trajectory tile");
  constexpr int TPW = MT::MLP ? (T64 ? 64 : (T1 ? 1 : (T4 ? 4 : 16 * NSETS))) : (RT > 0 ? RT : 64);
  constexpr int LPS = (MT::MLP && !T64) ? (T1 ? 1 : (T4 ? 4 : 16)) : TPW;
  static_assert(MT::MLP || G == 1, "closed-form models use one wavefront per tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wgw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wavefront inside the workgroup (wave-uniform: keeps tile guards scalar)
  const int wave = LW ? 0 : wgw;                                       // wavefront inside the TILE (lane-wise kernels: a tile is one wavefront)
  // lane-wise kernels: four tiles per workgroup.  Workgroups go round-robin over the 8 XCDs; tile t keeps landing on XCD t % 8 (the
  // protocol-major launch order deals the protocols out per XCD on that assumption): workgroup b, wavefront w -> tile (b % 8) + 8 (4 (b / 8) + w)
  const int tile = LW ? (int)((blockIdx.x & 7u) + 8u * ((blockIdx.x >> 3) * (unsigned)IONODE_LW_TILES_PER_WG + (unsigned)wgw)) : (int)blockIdx.x;
  unsigned char *const smem_t = LW ? smem + (size_t)wgw * (size_t)a.lw_bytes : smem;
  if constexpr (LW) {
    // the grid is rounded up to whole workgroups on every XCD: a tile past the batch leaves at once (it has no trajectory, and with
    // several weight images no image: its index would point past the caller's array).  No barrier joins the tiles of a lane-wise
    // workgroup after this point except MlpTile::init's, which the hardware completes without the wavefronts that have ended.
    if (tile * (MT::MLP ? 64 : (RT > 0 ? RT : 64)) >= a_in.B) return;
  }
  const int j = lane % LPS;
  const int cset = (NSETS > 1) ? wave / WPS : 0, wis = wave % WPS;  // column set of this wavefront, wavefront index inside the set
  const bool primary = (lane < LPS) && (wis == 0);  // the replica that writes per-trajectory scalars
  const int slot_raw = tile * TPW + cset * LPS + j;   // launch slot; the trajectory it integrates: a.order[slot] (or slot)
  const bool valid = slot_raw < a.B;
  const int slot_c = valid ? slot_raw : a.B - 1;
  const int traj = a.order ? a.order[slot_c] : slot_c;
  const int traj_raw = valid ? traj : a.B;   // (== 0 only for the lane that owns trajectory 0: the step log)

  // PD slot of a 64-per-wavefront MLP kernel: 1 = the MFMA form (any N <= 16), 10 = the per-lane vector-ALU net for N = 10 (MlpLane)
  constexpr bool VNET = T64 && PD > 1;
  // NT slot == 0 of a four-wavefront MLP kernel: the run-time-width tile (MlpGen: any N <= 512 without a tuned tile of its own)
  constexpr bool GEN = MT::MLP && G == 4 && NT == 0;
  using MlpTileT = MlpTile<G, (T64 ? 1 : (RT > 0 ? RT : 1)), (NT > 0 ? NT : 1), ((PD > 0 && !VNET) ? PD : 1), (NSETS > 1 ? 4 : 0)>;
  using MlpT = typename std::conditional<VNET, MlpLane<(VNET ? PD : 10)>, typename std::conditional<T1, typename std::conditional<(TAIL & 64) != 0, MlpRow1Deep, MlpRow1>::type, typename std::conditional<T4, MlpTile4, typename std::conditional<GEN, MlpGen, MlpTileT>::type>::type>::type>::type;
  typename std::conditional<MT::MLP, MlpT, NoMlp>::type mlp;
  if constexpr (MT::MLP) mlp.init(a, smem_t, wave, lane, tile * TPW);
  // lane-wise kernels: interpolant rows + tail buffers; behind the MlpTile region when there is one
  size_t lw_off = 0;
  if constexpr (T64) lw_off = (MlpT::lds_bytes(a.L) + 15) & ~(size_t)15;
  unsigned char *const lsm = smem_t + lw_off;
  STAMP_DECL
#ifdef IONODE_STAMPS
  if constexpr (MT::MLP) mlp.sp = &stamps_;
#endif

  double p[NPAR];   // (not const: made opaque once per attempt in the lane-wise kernels, below)
#pragma unroll
  for (int i = 0; i < NPAR; ++i) p[i] = a.params[(size_t)traj * a.n_params + i];
  const int pidx = a.prot_of_traj ? a.prot_of_traj[traj] : (traj % a.P);
  const double *__restrict__ pv = a.prot_v + (size_t)pidx * a.Np;

  const S rtol = (S)a.rtol, atol = (S)a.atol;
  S *__restrict__ yout = reinterpret_cast<S *>(a.y_out) + (size_t)traj * a.Nt * D;
  double *__restrict__ iout = a.i_out ? a.i_out + (size_t)traj * a.Nt : nullptr;
  const int Nt = a.Nt;

  S y[D], f[D];
#pragma unroll
  for (int d = 0; d < D; ++d) y[d] = reinterpret_cast<const S *>(a.y0)[(size_t)traj * D + d];

  double t = a.t_eval[0];
  {
    double v0;
    const bool in0 = protocol_v(a, pv, (double)(S)t, v0);
    rhs<MODEL, S, T64>(a, p, v0, in0, y, f, mlp);  // f0 = func(t[0], y0)
  }

  // _select_initial_step (order argument 4), all in the state dtype
  double dt;
  {
    const S t0s = (S)t;
    S scale[D], tmp[D], y1[D], f1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) scale[d] = atol + abs_(y[d]) * rtol;
#pragma unroll
    for (int d = 0; d < D; ++d) tmp[d] = y[d] / scale[d];
    const S d0 = rms_norm<S, D>(tmp);
#pragma unroll
    for (int d = 0; d < D; ++d) tmp[d] = f[d] / scale[d];
    const S d1 = rms_norm<S, D>(tmp);
    S h0;
    if (d0 < (S)1e-5 || d1 < (S)1e-5) h0 = (S)1e-6;
    else h0 = (S)0.01 * d0 / d1;
#pragma unroll
    for (int d = 0; d < D; ++d) y1[d] = y[d] + h0 * f[d];
    {
      double v1;
      const bool in1 = protocol_v(a, pv, (double)(t0s + h0), v1);
      rhs<MODEL, S, T64>(a, p, v1, in1, y1, f1, mlp);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) tmp[d] = (f1[d] - f[d]) / scale[d];
    const S d2 = rms_norm<S, D>(tmp) / h0;
    S h1;
    if (d1 <= (S)1e-15 && d2 <= (S)1e-15) {
      const S c = h0 * (S)1e-3;
      h1 = (S)1e-6 > c ? (S)1e-6 : c;
    } else {
      h1 = (S)det_root5((double)((S)0.01 / (d1 > d2 ? d1 : d2)));
    }
    const S h = ((S)100 * h0 < h1) ? (S)100 * h0 : h1;
    dt = (double)h;
    if (dt > a.dt_max) dt = a.dt_max;
  }

  // ---- lane-wise kernels: where the dense output goes through ----
  // gfx9 counts loads and stores in ONE in-order counter (vmcnt): a load issued behind a store cannot be consumed before that
  // store has been acknowledged by L2 -- for a store to a cold line that is an HBM round trip.  The round-1 emission loop
  // loaded t_eval once per 64-sample chunk behind the previous chunk's store, and the stage voltages of the next attempt
  // behind all of them: waves sat in s_waitcnt 82 % of the time (SQ_WAIT_ANY, profiles/r02_closed_form.md).  On a verified
  // uniform output grid (te_exact) output times are formed arithmetically and the next attempt's protocol lookups are issued
  // and consumed BEFORE the emission, so the emission is LDS reads + VALU + stores only and nothing waits on a store.
  // (Rounds 2-3 also held samples back in LDS tail buffers until a whole 64-byte sector could be written; L2 merges the partial
  // sectors of neighbouring steps on its own, and without the tail logic the lean 2-state kernel fits 128 registers and
  // 8.5 KiB of LDS -- four wavefronts per SIMD instead of three: 35.6 -> 30.5 ms at 393 216 x 20 001, profiles/r04_emission_ab.md.)
  constexpr bool CF2 = LW && D == 2;
  constexpr int LT = MT::MLP ? (TAIL == 1 ? 1 : 0) : TAIL;   // the LDS layout's variant key (MLP kernels at 64 per wavefront: lean or general)
  constexpr int ROWB = LwLds::rowb(D);
  // TAIL == 2 of a closed-form kernel: the current / objective epilogue reads V(t_k) from the pre-pass table a.v_tab (selected
  // by the dispatcher when ionode_desc.v_at_outputs is given); a compile-time variant so that neither variant carries the
  // other's code and registers
  constexpr bool VTAB = !MT::MLP && TAIL == 2;
  double *const ssep = reinterpret_cast<double *>(lsm + LwLds::aux_off(D));  // [64][8] partial sums of the fused objective (not in the lean variants)
  if constexpr (LW) {
    if (a.sse_out != nullptr) {
#pragma unroll
      for (int m = 0; m < 8; ++m) ssep[lane * 8 + m] = 0.0;
    }
  }
  // owp: protocol index of every lane's trajectory, trl: its trajectory index (read per lane group by the emission).  clist: the work list.
  int *const owp = reinterpret_cast<int *>(lsm + LwLds::owp_off(D, LT));
  int *const trl = reinterpret_cast<int *>(lsm + LwLds::trl_off(D, LT));
  unsigned short *const clist = reinterpret_cast<unsigned short *>(lsm + LwLds::clist_off(D, LT));
  if constexpr (LW) {
    if (lane < LPS) trl[lane] = traj, owp[lane] = pidx;
  }
  auto te_at = [&](int idx) -> double { return a.te_t0 + (double)idx * a.te_dt; };

  // solution[0] = y0
  double sse = 0.0;  // fused objective: this lane's trajectory (accumulated by the owner wavefront's replica lanes)
  if (valid && primary) {
    if (a.y_out) store_state<S, D>(yout, y);
    if (iout) {
      double v0;
      protocol_v(a, pv, t, v0);
      S gate;
      if (a.obs_open) gate = y[D - 1]; else gate = y[0] * y[1];
      if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
      iout[0] = (double)gate * (v0 - a.obs_e);
    }
  }
  if (a.sse_out != nullptr && valid) {
    double v0;
    protocol_v(a, pv, t, v0);
    S gate;
    if (a.obs_open) gate = y[D - 1]; else gate = y[0] * y[1];
    if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
    const double r0 = (double)gate * (v0 - a.obs_e) - a.sse_ref[(size_t)pidx * Nt];
    sse = r0 * r0;
  }

  int oi = 1;  // next output index
  int nacc = 0, nrej = 0;
  int since = 0;  // attempts since the last emitted output (torchdiffeq counts max_num_steps per _advance call)
  int status = IONODE_STATUS_OK;
  bool active = valid && Nt > 1;
  const S nan_s = (S)__builtin_nan("");

  // stage voltages of the coming attempt: pure functions of (t, dt).  Closed-form kernels carry them across iterations: they
  // are looked up for the NEXT attempt right after the controller, ahead of the emission's stores (see "where the dense output goes through" above)
  double vst[5];
  bool inst[5];
  auto lookup_stages = [&](double tt, double dd) {
    const S tts = (S)tt, dds = (S)dd, tt1s = (S)(tt + dd);
#ifndef IONODE_BATCH_LOOKUPS_ALL
#define IONODE_BATCH_LOOKUPS_ALL 0
#endif
#ifndef IONODE_BATCH_LOOKUPS_T64
#define IONODE_BATCH_LOOKUPS_T64 0
#endif
#ifndef IONODE_BATCH_LOOKUPS_CF
#define IONODE_BATCH_LOOKUPS_CF 1   // closed-form kernels (6-state at one wavefront per SIMD: -4 .. -10 %; 2-state: -2 .. -3.4 %; not the
                                    // 2-state table variant TAIL == 2: 173 instead of 151 registers = two wavefronts per SIMD instead of three)
#endif
    // (not for the N <= 16 kernel at 64 per wavefront: 251 -> 272 VGPRs = one wavefront per SIMD, 58 -> 87 ms; the 2-state kernel lost
    // 4 % with it while its build still hoisted constants and spilled)
    if (((MT::MLP && G > 1) || (IONODE_BATCH_LOOKUPS_CF && !MT::MLP && (D > 2 || TAIL != 2)) || (IONODE_BATCH_LOOKUPS_T64 && T64) || IONODE_BATCH_LOOKUPS_ALL) && a.prot_t == nullptr) {
      // uniform protocol grid: five indices, five 16-byte loads back to back, then the interpolations -- ONE memory round
      // trip per attempt (protocol_v() per stage time waited for each pair of samples in turn: 5 dependent round trips,
      // ~7 k cycles of the s00 attempt)
      double tq[5], lo[5], hi[5];
      int ix[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        tq[i] = (double)((i >= 4) ? R::prev_(tt1s) : tts + (S)kAlpha[i] * dds);
        inst[i] = protocol_index(a, tq[i], ix[i]);
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) { lo[i] = pv[ix[i] - 1]; hi[i] = pv[ix[i]]; }
#pragma unroll
      for (int i = 0; i < 5; ++i) vst[i] = inst[i] ? protocol_from(a, lo[i], hi[i], ix[i], tq[i]) : a.v_oob;
      return;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const S ti = (i >= 4) ? R::prev_(tt1s) : tts + (S)kAlpha[i] * dds;  // alpha == 1: Perturb.PREV (stages 4 and 5)
      inst[i] = protocol_v(a, pv, (double)ti, vst[i]);
    }
  };
#ifndef IONODE_CARRY_V_TINY16
#define IONODE_CARRY_V_TINY16 1   // the N <= 16 kernel at 16 trajectories per wavefront: next attempt's lookups before the emission (25.9 -> 23.5 ms at 65 536)
#endif
#ifndef IONODE_CARRY_V_MLP
#define IONODE_CARRY_V_MLP 0  // tried for the MLP kernels too: +1 % time (372.6 -> 376.2 ms same box), kept off
#endif
#ifndef IONODE_CARRY_V_M6
#define IONODE_CARRY_V_M6 1   // lean 6-state variant (224 registers, headroom for five more voltages): next attempt's lookups ahead of the emission's stores
#endif
  constexpr bool CARRY_V = (LW && D == 2) || (IONODE_CARRY_V_M6 && LW && D > 2 && LEAN) || (MT::MLP && (IONODE_CARRY_V_MLP || (IONODE_CARRY_V_TINY16 && G == 1)));  // (6-state: +50 % at 2 wavefronts per SIMD, no change at 1 per SIMD -- 38.0 vs 37.9 ms)
  if constexpr (CARRY_V) lookup_stages(t, dt);

  for (;;) {
    if constexpr (LW) {
      // Lane-wise kernels run at a fixed register budget (2-state: 168 for three wavefronts per SIMD).  Everything DERIVED from the
      // per-lane parameters that is invariant over the attempts -- fp32 copies and out-of-range rate constants for the fp32-state
      // rule, negated exponents, protocol row addresses -- would be hoisted out of this loop and stay live through it: ~20 VGPRs for
      // values the hot path never reads.  An empty asm makes the parameters opaque once per attempt: no instruction, no hoisting.
#pragma unroll
      for (int i = 0; i < NPAR; ++i) asm volatile("" : "+v"(p[i]));
    }
    // ---- per-trajectory assertions of _adaptive_step / _advance ----
    bool failed_now = false;
    if (active) {
      if ((int64_t)since >= a.max_steps || (int64_t)nacc + nrej >= a.max_total) { status = IONODE_STATUS_MAX_STEPS; failed_now = true; }
      else if (!(t + dt > t)) { status = IONODE_STATUS_DT_UNDERFLOW; failed_now = true; }
      else {
        bool fin = true;
#pragma unroll
        for (int d = 0; d < D; ++d) fin = fin && isfinite((double)y[d]);
        if (!fin) { status = IONODE_STATUS_NONFINITE; failed_now = true; }
      }
      if (failed_now) active = false;
    }
    // failed trajectories: the rest of their output is NaN (cooperative fill)
    {
      unsigned long long fm = __ballot(failed_now && lane < LPS);
      while (fm) {
        const int jj = __builtin_ctzll(fm);
        fm &= fm - 1;
        if (WPS == 1 || (jj % WPS) == wis) {
          const int o0 = __builtin_amdgcn_readlane(oi, jj);
          const int tr = __builtin_amdgcn_readlane(traj, jj);
          S *__restrict__ yo = reinterpret_cast<S *>(a.y_out) + (size_t)tr * Nt * D;
          for (int idx = o0 + lane; idx < Nt && (a.y_out || a.i_out); idx += 64) {
            if (a.y_out) {
#pragma unroll
              for (int d = 0; d < D; ++d) yo[(size_t)idx * D + d] = nan_s;
            }
            if (a.i_out) a.i_out[(size_t)tr * Nt + idx] = __builtin_nan("");
          }
        }
      }
    }
    if constexpr (NSETS > 1) {
      // two column sets: the wavefronts of one set may be done while the other set still integrates -- but every stage evaluation is a
      // collective of all G wavefronts (each computes its row tiles for BOTH sets), so the tile leaves the loop together
      if (__syncthreads_or(active ? 1 : 0) == 0) break;
    } else {
      if (__ballot(active) == 0ull) break;
    }

    // ---- _runge_kutta_step ----
    const double t0 = t;
    const double t1 = t0 + dt;
    const S dts = (S)dt;
    S k[7][D], yi[D];
#pragma unroll
    for (int d = 0; d < D; ++d) k[0][d] = f[d];
    // stage voltages: pure functions of (t0, dt), so all protocol loads are issued ahead of the stages
    if constexpr (!CARRY_V) lookup_stages(t0, dt);
    STAMP(stamps_, 1);  // slot 1 (asm tile: layer 0 is inside the stream): attempt prologue = stage-voltage lookups
    ClosedRates<MT::MLP ? IONODE_MODEL_HH2 : MODEL> cr;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      S bd[6];
#pragma unroll
      for (int jx = 0; jx <= i; ++jx) bd[jx] = (S)kBeta[i][jx] * dts;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        S s = k[0][d] * bd[0];
#pragma unroll
        for (int jx = 1; jx <= i; ++jx) s = s + k[jx][d] * bd[jx];
        yi[d] = y[d] + s;
      }
      if constexpr (MT::MLP && T64) {
        // the lane-wise nets: the rate constants of the Hodgkin-Huxley terms are reused as in the closed-form kernels below
        bool fresh = (i == 0);
        if (i > 0 && i < 5) fresh = __ballot(vst[i] != vst[i > 0 ? i - 1 : 0] || inst[i] != inst[i > 0 ? i - 1 : 0]) != 0ull;
        rhs<MODEL, S, T64>(a, p, vst[i < 4 ? i : 4], inst[i < 4 ? i : 4], yi, k[i + 1], mlp, &cr, fresh);
      } else if constexpr (MT::MLP) rhs<MODEL, S, T64>(a, p, vst[i < 4 ? i : 4], inst[i < 4 ? i : 4], yi, k[i + 1], mlp);
      else {
        // rate constants depend on the stage VOLTAGE only: i == 5 shares its stage time with i == 4, and on the holding / step
        // segments of the reference's protocols (Pr3, Pr5, staircase plateaus: train-s1.py:69-95) consecutive stages see the very same
        // voltage -- when every lane of the wavefront does, the previous stage's rates are reused (same inputs, same bits):
        // 4 instead of 20 exp per attempt of the 2-state model on a plateau, 12 instead of 60 for the 6-state model
        if (i < 5) {
          bool fresh = (i == 0);
          if (i > 0) fresh = __ballot(vst[i] != vst[i > 0 ? i - 1 : 0] || inst[i] != inst[i > 0 ? i - 1 : 0]) != 0ull;
          if (fresh) closed_rates<MODEL, S>(a, p, vst[i], inst[i], cr);
        }
        closed_rhs<MODEL, S>(cr, yi, k[i + 1]);
      }
    }
    // y1 = y_5 (c_sol == beta[5] + [0]); error estimate; _compute_error_ratio
    S tmp[D];
    {
      S be[7];
#pragma unroll
      for (int jx = 0; jx < 7; ++jx) be[jx] = dts * (S)kCerr[jx];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        S e = k[0][d] * be[0];
#pragma unroll
        for (int jx = 1; jx < 7; ++jx) e = e + k[jx][d] * be[jx];
        const S ay0 = abs_(y[d]), ay1 = abs_(yi[d]);
        const S tol = atol + rtol * (ay0 > ay1 ? ay0 : ay1);
        tmp[d] = e / tol;
      }
    }
    const S ratio = abs_(rms_norm<S, D>(tmp));
    const bool accept = ratio <= (S)1;

    // _optimal_step_size (fp64)
    double dt_next;
    if (ratio == (S)0) dt_next = dt * 10.0;
    else {
      const double dfactor = (ratio < (S)1) ? 1.0 : 0.2;
      const double er = (double)ratio;
      double fac = 0.9 / det_root5(er);
      if (!(fac > dfactor)) fac = dfactor;
      if (!(fac < 10.0)) fac = 10.0;
      if (er != er) fac = __builtin_nan("");
      dt_next = dt * fac;
    }

    STAMP(stamps_, 6);  // slot 6: stage assembly + error control (scalar RK work outside the MLP)
    const bool acc_now = active && accept;
#ifndef IONODE_STAMPS
    if (a.step_log != nullptr && active && primary && traj_raw == 0 && (int64_t)nacc + nrej < a.step_log_cap) {
      double *row = a.step_log + 4 * ((int64_t)nacc + nrej);
      row[0] = t0; row[1] = dt; row[2] = (double)ratio; row[3] = accept ? 1.0 : 0.0;
    }
#endif
    const int nacc_before = nacc, oi_before = oi;
    if (active) { if (accept) ++nacc; else ++nrej; }
    const double dt_capped = (dt_next > a.dt_max) ? a.dt_max : dt_next;  // NaN stays NaN (-> 'underflow in dt')
    if constexpr (CARRY_V) lookup_stages(acc_now ? t1 : t0, (active || acc_now) ? dt_capped : dt);

    // ---- _interp_fit + cooperative dense output ----
    // x = (t_k - t0) / (t1 - t0) of every dense-output sample: ONE division per attempt (the reciprocal of the step length),
    // then div_by() per sample -- the same correctly rounded quotient
    const double den = t1 - t0;
    const double rden = 1.0 / den;
    constexpr int ROW = 4 + 5 * D;  // doubles per LDS row, 16-byte aligned rows
    S ic[LW ? 1 : 5][LW ? 1 : D];   // e, d, c, b, a -- tile kernels keep them in registers (broadcast by v_readlane)
    {
      S bm[7];
#pragma unroll
      for (int jx = 0; jx < 7; ++jx) bm[jx] = dts * (S)kCmid[jx];
      auto fit = [&](int d, S *c5) {
        S s = k[0][d] * bm[0];
#pragma unroll
        for (int jx = 1; jx < 7; ++jx) s = s + k[jx][d] * bm[jx];
        const S YM = y[d] + s;
        const S F0 = k[0][d], F1 = k[6][d], Y0 = y[d], Y1 = yi[d];
        c5[4] = ((S)2 * dts) * (F1 - F0) - (S)8 * (Y1 + Y0) + (S)16 * YM;
        c5[3] = dts * ((S)5 * F0 - (S)3 * F1) + (S)18 * Y0 + (S)14 * Y1 - (S)32 * YM;
        c5[2] = dts * (F1 - (S)4 * F0) - (S)11 * Y0 - (S)5 * Y1 + (S)16 * YM;
        c5[1] = dts * F0;
        c5[0] = Y0;
      };
      if constexpr (LW) {
        // Lane-wise kernels: a lane's interpolant (t0, step length, its reciprocal, 5 x D coefficients) goes to its LDS row;
        // the wavefront then reads the emitting trajectory's row at a uniform address (7 broadcast ds_read_b128 for D = 2)
        // instead of ~29 v_readlane per emitting trajectory.  The workgroup is one wavefront: LDS is in order, no barrier.
        // The coefficients are fitted and stored two components at a time, so that at most 10 of the 5 x D are live
        // (6-state model: 60 registers fewer at the kernel's pressure peak).
        double2 *row = reinterpret_cast<double2 *>(lsm + lane * ROWB);
        row[0] = make_double2(t0, den);
        row[1] = make_double2(rden, 0.0);
        static_assert(D % 2 == 0, "rows hold component pairs");
#pragma unroll
        for (int d = 0; d < D; d += 2) {
          S ca[5], cb2[5];
          fit(d, ca);
          fit(d + 1, cb2);
#pragma unroll
          for (int c = 0; c < 5; ++c) row[2 + (c * D + d) / 2] = make_double2((double)ca[c], (double)cb2[c]);
          if constexpr (D > 2) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          S c5[5];
          fit(d, c5);
#pragma unroll
          for (int c = 0; c < 5; ++c) ic[c][d] = c5[c];
        }
      }
    }
    STAMP(stamps_, 8);  // slot 8: interpolant fit
    if (LEAN || LEANM || a.te_dt > 0.0) {
      // ---- output cursor, lane-parallel: how many requested times fall in (t0, t1] for MY trajectory? ----
      // Guess the last index from the (nearly) uniform output grid, then VERIFY against t_eval itself and walk to the
      // exact answer: correct for any increasing t_eval, one L2 round trip for the whole tile when the guess is right
      // (instead of one dependent load per trajectory in the cooperative scan below).
      int n_out = 0;
      if (acc_now) {
        const double gf = floor((t1 - a.te_t0) * a.te_rdt);  // a guess: verified below
        long long g = (gf < (double)(oi - 1)) ? (long long)(oi - 1) : ((gf > (double)(Nt - 1)) ? (long long)(Nt - 1) : (long long)gf);
        // lean lane-wise kernels: the 6-state one verifies its cursor against arithmetic times too (262 144 x 20 001: 76.1 -> 72.1 ms);
        // the 2-state ones keep the load -- the two scalars cost them 24 spilled SGPRs (393 216: 31.1 -> 38.0 ms)
        if ((LW && LEAN && D > 2) || (!LW && a.te_exact)) {  // verified uniform output grid: t_k is formed arithmetically, no load
          while (g >= oi && te_at((int)g) > t1) --g;
          while (g + 1 < Nt && te_at((int)g + 1) <= t1) ++g;
        } else {
          while (g >= oi && a.t_eval[g] > t1) --g;
          while (g + 1 < Nt && a.t_eval[g + 1] <= t1) ++g;
        }
        n_out = (int)(g - oi + 1);
      }
      STAMP(stamps_, 9);  // slot 9: output cursor
      if constexpr (!LW && G > 1) {
      // ---- MLP tile kernels: the owner wavefront emits its NS = TPW / G trajectories.  Everything that has to come from
      // memory for the first 64-sample chunk of ALL of them -- output times (unless the grid is verified uniform: arithmetic),
      // the two protocol samples per output time of the observation model -- is issued before anything is evaluated: one
      // round trip per accepted step instead of two dependent ones per emitting trajectory.  Samples beyond the first chunk
      // (steps spanning more than 64 outputs) take the plain loop.
      constexpr int NS = (LPS >= WPS) ? LPS / WPS : 1;   // (one trajectory per tile: wavefront 0 emits it, the others nothing)
      const bool exact = a.te_exact != 0;
      const bool want_i = (a.i_out != nullptr) || (a.sse_out != nullptr);
      const bool ugrid = a.prot_t == nullptr;
      int o_[NS], n_[NS], ip_[NS];
      double tk_[NS], plo_[NS], phi_[NS];
      bool inr_[NS];
      const double *pv_[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int jj = wis + WPS * k;
        o_[k] = __builtin_amdgcn_readlane(oi, jj);
        n_[k] = (LPS >= WPS || jj < LPS) ? __builtin_amdgcn_readlane(n_out, jj) : 0;
        pv_[k] = a.prot_v + (size_t)__builtin_amdgcn_readlane(pidx, jj) * a.Np;
        tk_[k] = 0.0;
        if (lane < n_[k]) tk_[k] = exact ? te_at(o_[k] + lane) : a.t_eval[o_[k] + lane];
      }
      if (want_i && ugrid) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          inr_[k] = protocol_index(a, tk_[k], ip_[k]);  // (idle lanes: t = 0, a valid index; their loads are harmless)
          plo_[k] = pv_[k][ip_[k] - 1]; phi_[k] = pv_[k][ip_[k]];
        }
      }
      STAMP(stamps_, 10);  // slot 10: emission, gather phase
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const int jj = wis + WPS * k;
        const int n = n_[k], o = o_[k];
        if (n > 0) {
          const double t0b = bcast_f64(t0, jj), denb = bcast_f64(den, jj), rdenb = bcast_f64(rden, jj);
          S cb[5][D];
#pragma unroll
          for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < D; ++d) cb[c][d] = bcast<S>(ic[c][d], jj);
          const int tr = __builtin_amdgcn_readlane(traj, jj);
          S *__restrict__ yo = a.y_out ? reinterpret_cast<S *>(a.y_out) + (size_t)tr * Nt * D : nullptr;
          double *__restrict__ io = a.i_out ? a.i_out + (size_t)tr * Nt : nullptr;
          const double *__restrict__ refb = a.sse_out ? a.sse_ref + (size_t)__builtin_amdgcn_readlane(pidx, jj) * Nt : nullptr;
          double sacc = 0.0;
          for (int c0 = 0; c0 < n; c0 += 64) {
            const int idx = o + c0 + lane;
            double tk = tk_[k];
            if (c0 > 0 && c0 + lane < n) tk = exact ? te_at(idx) : a.t_eval[idx];
            if (c0 + lane < n) {
              const S x = (S)div_pos(tk - t0b, denb, rdenb);  // _interp_evaluate: x = (t - t0) / (t1 - t0) in fp64, cast; running powers
              S out[D];
              S xp = x;
#pragma unroll
              for (int d = 0; d < D; ++d) out[d] = cb[0][d] + x * cb[1][d];
#pragma unroll
              for (int c = 2; c < 5; ++c) {
                xp = xp * x;
#pragma unroll
                for (int d = 0; d < D; ++d) out[d] = out[d] + xp * cb[c][d];
              }
              if (yo) store_state<S, D>(yo + (size_t)idx * D, out);
              if (want_i) {
                double vk;
                if (c0 == 0 && ugrid) vk = inr_[k] ? protocol_from(a, plo_[k], phi_[k], ip_[k], tk) : a.v_oob;
                else protocol_v(a, pv_[k], tk, vk);
                S gate;
                if (a.obs_open) gate = out[D - 1]; else gate = out[0] * out[1];
                if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
                const double ik = (double)gate * (vk - a.obs_e);
                if (io) io[idx] = ik;
                if (refb) { const double rr = ik - refb[idx]; sacc += rr * rr; }
              }
            }
          }
          if (a.sse_out) {  // fused objective: the step's squared residuals of trajectory jj
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) sacc += __shfl_xor(sacc, msk);
            if (j == jj) sse += sacc;
          }
        }
      }
      oi += n_out;
      } else {
      // ---- lane-wise kernels on a verified uniform output grid: WORK-LIST emission.
      // Steps differ wildly in the number of output samples they cover (2-state, sine-wave legs: 10 % of the accepted steps cover
      // <= 5 samples, the median 39, 10 % >= 170; 6-state: ~20 on average).  Round 3 handed each group of 8 lanes one emitting
      // trajectory at a time and ran a pass until the longest of its 8 trajectories was done: 78 iterations per attempt where 36 would
      // do (a CPU replay of the step logs of one wavefront, tools/emit_replay.py), and the dense output was 72 % of the kernel.  Every emitting lane
      // appends its step's 8-sample chunks {lane, 8 * chunk number} to the LDS work list; a pass takes the next 8 entries, one per
      // group of 8 lanes; what used to be wave-uniform per trajectory (interpolant row, cursor, protocol, trajectory index) is read
      // per lane from LDS: the owner's row, whose spare slot carries (oi, n_out), the protocol index parked in `owp`, the
      // trajectory index in `trl`.  Same samples, same arithmetic; the fused objective keeps its summation order (a chunk = the same
      // 8 consecutive samples as before, partial sum number = chunk number -- which is why steps of more than 64 samples take the
      // one-trajectory-per-pass loop below); V(t_k) and the reference current of the NEXT pass are loaded before this pass's stores.
      bool packed_lane = false;   // my trajectory's samples are emitted by the work-list passes (the others: the loop below)
      if constexpr (LW && (VTAB || D > 2 || (CF2 && TAIL == 1))) {
        // a chunk of PK = 8 samples is served by PKL lanes x NSL samples each (lane kk: samples kk, kk + PKL, ...): one row read per
        // NSL samples -- with one sample per lane the LDS pipe, not the vector ALU, bounded these passes (the 6-state row is 272 bytes)
#ifndef IONODE_PACK_PKL_D6
#define IONODE_PACK_PKL_D6 2
#endif
        constexpr int PK = 8, PKL = (D == 2) ? 4 : IONODE_PACK_PKL_D6, NSL = PK / PKL;
        typedef S SV __attribute__((ext_vector_type(NSL)));
        const bool want_i = (a.i_out != nullptr) || (a.sse_out != nullptr);
        // one instance per compiled variant: the table variant (TAIL == 2) serves the current / objective epilogue, the plain one
        // states only (its epilogue without the table -- a protocol lookup per sample -- stays on the loop below)
        // (2-state kernels that also store the states keep the loop below: at ~34 samples per step its 64 consecutive samples per
        // store instruction touch half the cache lines of 8 x 8, and that path is store-bound: 41.5 against 44.7 ms packed)
        if (a.te_exact && (VTAB ? (want_i && (D > 2 || a.y_out == nullptr)) : (!want_i && a.y_out != nullptr))) {
          int lane_e = lane;   // opaque per-attempt copy: what is derived from it (row / list addresses, group and sample numbers) is computed here, per
                               // attempt, instead of being hoisted out of the attempt loop into a dozen VGPRs that stay live through the stage loop
          asm volatile("" : "+v"(lane_e));
          packed_lane = n_out > 0 && lane_e < LPS && n_out <= 64;
          const unsigned long long emd = __ballot(packed_lane);
          auto emit_packed = [&](auto wi_tag) {
            constexpr bool WI = decltype(wi_tag)::value;
            const int nch = (n_out + PK - 1) / PK;   // 1 .. 8 for the listed lanes
            if (packed_lane) *reinterpret_cast<int2 *>(lsm + lane_e * ROWB + 24) = make_int2(oi, oi + n_out);   // the row's spare slot
            const int x = nch - 1;
            const unsigned long long m0 = __ballot(packed_lane && (x & 1)), m1 = __ballot(packed_lane && (x & 2)), m2 = __ballot(packed_lane && (x & 4));
            const int q = mbcnt(m0, mbcnt(emd)) + 2 * mbcnt(m1) + 4 * mbcnt(m2);
            const int C = __builtin_popcountll(emd) + __builtin_popcountll(m0) + 2 * __builtin_popcountll(m1) + 4 * __builtin_popcountll(m2);
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (packed_lane && i < nch) clist[q + i] = (unsigned short)(lane_e | (i * PK) << 6);
            const int slot = lane_e / PKL, kk = lane_e % PKL;
            // decode of a list entry: trajectory lane, first sample of the lane, and (table variant) the loads of V(t_k) / reference
            struct Ent { int jj, idx0, end, part; bool has; double vk[NSL], rf[NSL]; };
            auto decode = [&](int c0) {
              Ent t;
              t.has = c0 + slot < C;
              const unsigned e = clist[t.has ? c0 + slot : (c0 < C ? c0 : 0)];
              t.jj = (int)(e & 63u);
              t.part = (int)(e >> 9);   // chunk number: the objective's partial-sum slot
              const int2 on2 = *reinterpret_cast<const int2 *>(lsm + t.jj * ROWB + 24);
              t.idx0 = on2.x + (int)(e >> 6) + kk;
              t.end = on2.y;
#pragma unroll
              for (int u = 0; u < NSL; ++u) { t.vk[u] = 0.0; t.rf[u] = 0.0; }
              if constexpr (VTAB && WI) {
                const int pj = owp[t.jj];
#pragma unroll
                for (int u = 0; u < NSL; ++u) {
                  const int idx = t.idx0 + u * PKL;
                  if (t.has && idx < t.end) {
                    t.vk[u] = a.v_tab[(size_t)pj * Nt + idx];
                    if (a.sse_out) t.rf[u] = a.sse_ref[(size_t)pj * Nt + idx];
                  }
                }
              }
              return t;
            };
            // 6-state kernels (one wavefront per SIMD, registers to spare): the next pass's entry and table loads are issued before this
            // pass is evaluated.  2-state kernels (three per SIMD, 168 registers): no cross-pass prefetch -- two live entries cost
            // 14-28 spilled registers, and the fused objective stores nothing its loads could queue behind
            constexpr bool PREF = (D > 2);
            Ent nx{};
            if constexpr (PREF) nx = decode(0);
            for (int c0 = 0; c0 < C; c0 += 64 / PKL) {
              const Ent cur = PREF ? nx : decode(c0);
              if constexpr (PREF) { if (c0 + 64 / PKL < C) nx = decode(c0 + 64 / PKL); }
              const int jj = cur.jj;
              const double2 *rj = reinterpret_cast<const double2 *>(lsm + jj * ROWB);
              const double2 h0 = rj[0];
              const double t0b = h0.x, denb = h0.y, rdenb = rj[1].x;
              S cb[5][D];
#pragma unroll
              for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int d = 0; d < D; d += 2) {
                  const double2 cc = rj[2 + (c * D + d) / 2];
                  cb[c][d] = (S)cc.x; cb[c][d + 1] = (S)cc.y;
                }
              const int tr = trl[jj];
              double tk[NSL];
              SV xv;
#pragma unroll
              for (int u = 0; u < NSL; ++u) {
                tk[u] = te_at(cur.idx0 + u * PKL);
                xv[u] = (S)div_pos(tk[u] - t0b, denb, rdenb);  // _interp_evaluate: x in fp64, cast; running powers
              }
              SV ov[D], xp = xv;
#pragma unroll
              for (int d = 0; d < D; ++d) ov[d] = cb[0][d] + xv * cb[1][d];
#pragma unroll
              for (int c = 2; c < 5; ++c) {
                xp = xp * xv;
#pragma unroll
                for (int d = 0; d < D; ++d) ov[d] = ov[d] + xp * cb[c][d];
              }
              double rr2[NSL];
#pragma unroll
              for (int u = 0; u < NSL; ++u) {
                const int idx = cur.idx0 + u * PKL;
                rr2[u] = 0.0;
                if (cur.has && idx < cur.end) {
                  S out[D];
#pragma unroll
                  for (int d = 0; d < D; ++d) out[d] = ov[d][u];
                  if (!WI || a.y_out) store_state<S, D>(reinterpret_cast<S *>(a.y_out) + ((size_t)tr * Nt + idx) * D, out);
                  if constexpr (WI) {
                    double vk;
                    if constexpr (VTAB) vk = cur.vk[u];  // == protocol_v(a, pvb, t_eval[idx]), evaluated once per protocol by the pre-pass
                    else protocol_v(a, a.prot_v + (size_t)owp[jj] * a.Np, tk[u], vk);
                    S gate;
                    if (a.obs_open) gate = out[D - 1]; else gate = out[0] * out[1];
                    if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
                    const double ik = (double)gate * (vk - a.obs_e);
                    if (a.i_out) a.i_out[(size_t)tr * Nt + idx] = ik;
                    if (a.sse_out) { const double rr = ik - (VTAB ? cur.rf[u] : a.sse_ref[(size_t)owp[jj] * Nt + idx]); rr2[u] = rr * rr; }
                  }
                }
              }
              if (WI && a.sse_out) {
                // the chunk's sum in the canonical tree ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) over its 8 consecutive samples
                double g8;
                if constexpr (PKL == 4) {
                  double ql = rr2[0] + dpp_f64<0xB1, 0xf>(rr2[0]), qh = rr2[1] + dpp_f64<0xB1, 0xf>(rr2[1]);   // quad_perm [1,0,3,2]
                  ql = ql + dpp_f64<0x4E, 0xf>(ql); qh = qh + dpp_f64<0x4E, 0xf>(qh);                           // quad_perm [2,3,0,1]
                  g8 = ql + qh;
                } else {
                  double sj[NSL];
#pragma unroll
                  for (int u = 0; u < NSL; ++u) sj[u] = rr2[u] + dpp_f64<0xB1, 0xf>(rr2[u]);
                  g8 = (sj[0] + sj[1]) + (sj[2] + sj[3]);
                }
                if (kk == 0 && cur.has) ssep[jj * 8 + cur.part] += g8;
              }
            }
          };
          if (emd) emit_packed(std::integral_constant<bool, VTAB>{});
        }
      }
      {
      // ---- owner wavefront evaluates and stores; the t_eval loads of the next trajectory are issued ahead ----
      unsigned long long em = __ballot(n_out > 0 && lane < LPS && !packed_lane);
      if (G > 1) {  // trajectory jj belongs to wavefront jj % G
        unsigned long long mine = 0ull;
#pragma unroll
        for (int k = 0; k < (TPW + G - 1) / G; ++k) mine |= 1ull << (wave + k * G);   // (fewer trajectories than wavefronts: masked by lane < LPS above)
        em &= mine;
      }
      int jj = em ? __builtin_ctzll(em) : 0;
      int o = __builtin_amdgcn_readlane(oi, jj);
      // closed-form kernels on a verified uniform output grid form t_k arithmetically (bit-equal to the t_eval entry)
      const bool arith_t = LW && a.te_exact;
      double tk_nxt = (em && o + lane < Nt) ? (arith_t ? te_at(o + lane) : a.t_eval[o + lane]) : 0.0;
      // table variant: V(t_k) and the reference current of the next trajectory's first chunk are in flight as well
      double vk_nxt = 0.0, rf_nxt = 0.0;
      auto prefetch_obs = [&](int jx, int ox) {
        if constexpr (VTAB) {
          const int pjx = __builtin_amdgcn_readlane(pidx, jx);
          if (ox + lane < Nt) {
            vk_nxt = a.v_tab[(size_t)pjx * Nt + ox + lane];
            if (a.sse_out) rf_nxt = a.sse_ref[(size_t)pjx * Nt + ox + lane];
          }
        }
      };
      if (em) prefetch_obs(jj, o);
      while (em) {
        em &= em - 1;
        const int jn = em ? __builtin_ctzll(em) : 0;
        const int on = __builtin_amdgcn_readlane(oi, jn);
        double tk = tk_nxt;
        const double vk_first = vk_nxt, rf_first = rf_nxt;
        if (em && on + lane < Nt) tk_nxt = arith_t ? te_at(on + lane) : a.t_eval[on + lane];  // next trajectory's first chunk, in flight meanwhile
        if (em) prefetch_obs(jn, on);
        const int n = __builtin_amdgcn_readlane(n_out, jj);
        double t0b, denb, rdenb;
        S cb[5][D];
        if constexpr (LW) {
          const double2 *rj = reinterpret_cast<const double2 *>(lsm) + jj * (ROW / 2);
          const double2 h0 = rj[0], h1 = rj[1];
          t0b = h0.x; denb = h0.y; rdenb = h1.x;
#pragma unroll
          for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < D; d += 2) {
              const double2 cc = rj[2 + (c * D + d) / 2];
              cb[c][d] = (S)cc.x; cb[c][d + 1] = (S)cc.y;
            }
        } else {
          t0b = bcast_f64(t0, jj); denb = bcast_f64(den, jj); rdenb = bcast_f64(rden, jj);
#pragma unroll
          for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < D; ++d) cb[c][d] = bcast<S>(ic[c][d], jj);
        }
        const int tr = __builtin_amdgcn_readlane(traj, jj);
        S *__restrict__ yo = a.y_out ? reinterpret_cast<S *>(a.y_out) + (size_t)tr * Nt * D : nullptr;
        double *__restrict__ io = nullptr;
        const double *__restrict__ pvb = nullptr, *__restrict__ refb = nullptr, *__restrict__ vtb = nullptr;
        if (a.i_out || a.sse_out) {
          if (a.i_out) io = a.i_out + (size_t)tr * Nt;
          int pj;
          if constexpr (LW) pj = __builtin_amdgcn_readlane(pidx, jj);  // no dependent global load per emitting trajectory
          else pj = a.prot_of_traj ? a.prot_of_traj[tr] : (tr % a.P);
          pvb = a.prot_v + (size_t)pj * a.Np;
          if (a.sse_out) refb = a.sse_ref + (size_t)pj * Nt;
          if constexpr (VTAB) vtb = a.v_tab + (size_t)pj * Nt;
        }
        double sacc = 0.0;
        for (int c0 = 0; c0 < n; c0 += 64) {
          const int idx = o + c0 + lane;
          if (c0 > 0 && c0 + lane < n) tk = arith_t ? te_at(idx) : a.t_eval[idx];
          if (c0 + lane < n) {
            const S x = (S)div_pos(tk - t0b, denb, rdenb);  // _interp_evaluate: x = (t - t0) / (t1 - t0) in fp64, cast; running powers
            S out[D];
            S xp = x;
#pragma unroll
            for (int d = 0; d < D; ++d) out[d] = cb[0][d] + x * cb[1][d];
#pragma unroll
            for (int c = 2; c < 5; ++c) {
              xp = xp * x;
#pragma unroll
              for (int d = 0; d < D; ++d) out[d] = out[d] + xp * cb[c][d];
            }
            if (yo) store_state<S, D>(yo + (size_t)idx * D, out);
            if (pvb) {
              double vk;
              if constexpr (VTAB) vk = (c0 == 0) ? vk_first : vtb[idx];  // == protocol_v(a, pvb, t_eval[idx]), evaluated once per protocol by the pre-pass
              else protocol_v(a, pvb, tk, vk);
              S gate;
              if (a.obs_open) gate = out[D - 1]; else gate = out[0] * out[1];
              if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
              const double ik = (double)gate * (vk - a.obs_e);
              if (io) io[idx] = ik;
              if (refb) { const double rr = ik - ((VTAB && c0 == 0) ? rf_first : refb[idx]); sacc += rr * rr; }
            }
          }
        }
        if (a.sse_out) {  // fused objective: the step's squared residuals of trajectory jj
          if constexpr (LW) {
            // lane-wise kernels: a full wavefront reduction per emitting trajectory (6 DPP steps + broadcast) was a quarter of
            // the epilogue's instructions.  Reduce over groups of 8 lanes only and keep 8 partial sums per trajectory in LDS
            // (the aux region); they are added up once, at the end.
            const double g8 = group8_sum_f64(sacc);
            if ((lane & 7) == 0) ssep[jj * 8 + (lane >> 3)] += g8;
          } else {
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) sacc += __shfl_xor(sacc, msk);
            if (j == jj) sse += sacc;
          }
        }
        jj = jn;
        o = on;
      }
      oi += n_out;
      }
      }
    } else {
      // ---- no grid hint: cooperative scan, every wavefront advances every cursor ----
      unsigned long long em = __ballot(acc_now && lane < LPS);
      while (em) {
        const int jj = __builtin_ctzll(em);
        em &= em - 1;
        const bool owner = (WPS == 1) || ((jj % WPS) == wis);
        int o = __builtin_amdgcn_readlane(oi, jj);
        const double t1b = bcast_f64(t1, jj);
        // every wavefront advances the output cursor; only the owner evaluates and stores
        double t0b = 0.0, denb = 1.0, rdenb = 1.0;
        S cb[5][D];
        S *__restrict__ yo = nullptr;
        double *__restrict__ io = nullptr;
        const double *__restrict__ pvb = nullptr;
        if (owner) {
          if constexpr (LW) {
            const double2 *rj = reinterpret_cast<const double2 *>(lsm) + jj * (ROW / 2);
            const double2 h0 = rj[0], h1 = rj[1];
            t0b = h0.x; denb = h0.y; rdenb = h1.x;
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
              for (int d = 0; d < D; d += 2) {
                const double2 cc = rj[2 + (c * D + d) / 2];
                cb[c][d] = (S)cc.x; cb[c][d + 1] = (S)cc.y;
              }
          } else {
            t0b = bcast_f64(t0, jj); denb = bcast_f64(den, jj); rdenb = bcast_f64(rden, jj);
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
              for (int d = 0; d < D; ++d) cb[c][d] = bcast<S>(ic[c][d], jj);
          }
          const int tr = __builtin_amdgcn_readlane(traj, jj);
          yo = reinterpret_cast<S *>(a.y_out) + (size_t)tr * Nt * D;
          if (a.i_out) {
            io = a.i_out + (size_t)tr * Nt;
            const int pj = a.prot_of_traj ? a.prot_of_traj[tr] : (tr % a.P);
            pvb = a.prot_v + (size_t)pj * a.Np;
          }
        }
        for (;;) {
          const int idx = o + lane;
          const double tk = (idx < Nt) ? a.t_eval[idx] : __builtin_inf();
          const bool ok = tk <= t1b;
          if (owner && ok) {
            // _interp_evaluate: x in fp64, cast; running powers
            const S x = (S)div_pos(tk - t0b, denb, rdenb);
            S out[D];
            S xp = x;
#pragma unroll
            for (int d = 0; d < D; ++d) out[d] = cb[0][d] + x * cb[1][d];
#pragma unroll
            for (int c = 2; c < 5; ++c) {
              xp = xp * x;
#pragma unroll
              for (int d = 0; d < D; ++d) out[d] = out[d] + xp * cb[c][d];
            }
            store_state<S, D>(yo + (size_t)idx * D, out);
            if (io) {
              double vk;
              protocol_v(a, pvb, tk, vk);
              S gate;
              if (a.obs_open) gate = out[D - 1]; else gate = out[0] * out[1];
              if (a.obs_g != 1.0) gate = (S)a.obs_g * gate;
              io[idx] = (double)gate * (vk - a.obs_e);
            }
          }
          const int n = __builtin_popcountll(__ballot(ok));
          o += n;
          if (n < 64) break;
        }
        if (j == jj) oi = o;
      }
    }
    STAMP(stamps_, 7);  // slot 7: interpolant fit + cooperative dense output
    if (active) since = (acc_now && oi > oi_before) ? 0 : since + 1;
    // ---- checkpoint of the accepted step for the backward sweep: (t0, dt, first output index, outputs, y, k1..k7) ----
    if (a.ckpt != nullptr && acc_now && primary && nacc_before < a.ckpt_cap) {
      double *__restrict__ rec = a.ckpt + ((size_t)traj * a.ckpt_cap + nacc_before) * (4 + 8 * D);
      rec[0] = t0; rec[1] = dt; rec[2] = (double)oi_before; rec[3] = (double)(oi - oi_before);
#pragma unroll
      for (int d = 0; d < D; ++d) rec[4 + d] = (double)y[d];
#pragma unroll
      for (int jx = 0; jx < 7; ++jx)
#pragma unroll
        for (int d = 0; d < D; ++d) rec[4 + D + jx * D + d] = (double)k[jx][d];
    }
    // ---- advance the RK state ----
    if (acc_now) {
#pragma unroll
      for (int d = 0; d < D; ++d) { y[d] = yi[d]; f[d] = k[6][d]; }
      t = t1;
      if (oi >= Nt) active = false;  // all requested outputs produced
    }
    if (active || acc_now) dt = dt_capped;
  }

#ifdef IONODE_STAMPS
  STAMP(stamps_, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.step_log != nullptr && a.step_log_cap >= 4)
    for (int i_ = 0; i_ < 16; ++i_) a.step_log[i_] = (double)stamps_.acc[i_];
#ifdef IONODE_ASM_STAMPS  // per-position cycle sums of the asm stream (tools/gen_mlp_asm.py --stamps): lanes of a[92]
  if constexpr (MT::MLP && G == 4 && NT == 13) {
    unsigned sv_;
    asm volatile("v_accvgpr_read_b32 %0, a92" : "=v"(sv_));
    if (blockIdx.x == 0 && threadIdx.x < 16 && a.step_log != nullptr && a.step_log_cap >= 8) a.step_log[16 + threadIdx.x] = (double)sv_;
  }
#endif
#endif
  if constexpr (LW) {
    if (a.sse_out != nullptr) {
#pragma unroll
      for (int m = 0; m < 8; ++m) sse += ssep[j * 8 + m];
    }
  }
  if (a.sse_out != nullptr && valid && lane < LPS && (WPS == 1 || (lane % WPS) == wis))
    a.sse_out[traj] = (status == IONODE_STATUS_OK) ? sse : __builtin_inf();  // the reference's time-limit rule: inf (train-d0.py:430-431)
  if (valid && primary) {
    a.status[traj] = status;
    if (a.stats) {
      int64_t *st = a.stats + (size_t)traj * 4;
      st[0] = nacc;
      st[1] = nrej;
      st[2] = 2 + 6 * ((int64_t)nacc + nrej);
      st[3] = status;
    }
  }
}

}  // namespace ionode
