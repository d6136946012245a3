// Shared device/host helpers for the gfx950 AR-DAE-VAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace ardae {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------------------------
// error convention of the C ABI: 0 ok, <0 invalid argument, >0 hipError_t
// ----------------------------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);

#define ARDAE_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      ::ardae::set_last_error(__VA_ARGS__);        \
      return -1;                                   \
    }                                              \
  } while (0)

#define ARDAE_HIP(call)                                                                  \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::ardae::set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (int)e__;                                                                   \
    }                                                                                    \
  } while (0)

#define ARDAE_LAUNCH_CHECK() ARDAE_HIP(hipGetLastError())

#define ARDAE_TRY(call)        \
  do {                         \
    int rc__ = (call);         \
    if (rc__ != 0) return rc__; \
  } while (0)

// Test / experiment switches (kernel selection, launch geometry): read from the environment ONLY when ARDAE_DEBUG_KNOBS=1 is
// set as well, so that a stray ARDAE_* variable cannot change what a production process runs.
inline const char* debug_knob(const char* name) {
  static const bool armed = [] { const char* e = getenv("ARDAE_DEBUG_KNOBS"); return e && e[0] == '1' && e[1] == 0; }();
  return armed ? getenv(name) : nullptr;
}

// ----------------------------------------------------------------------------------------------
// activations (reference: utils/models.py:14-32; F.softplus beta=1 threshold=20, F.relu)
// The saved tensor is always the POST-activation value a = act(pre); first/second derivatives are
// rebuilt from it:  softplus: s = sigmoid(pre) = 1 - exp(-a),  s' = s(1-s);  relu: s = [a>0], s' = 0.
// ----------------------------------------------------------------------------------------------
// get_nonlinear_func (utils/models.py:14-32): relu, softplus (csoftplus = log(exp(x) + 1) is the same function, evaluated in the accurate form), elu (alpha 1), tanh,
// leaky_relu (slope 0.2), swish (utils/models.py:8-10: x sigmoid(x); see swish_f below for how its derivatives are rebuilt from the
// saved output).  The software-pipelined N-row kernels are instantiated for NONE / RELU / SOFTPLUS (every shipped recipe); the other
// four run on the generic kernels.
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SOFTPLUS = 2, ACT_ELU = 3, ACT_TANH = 4, ACT_LEAKY = 5, ACT_SWISH = 6 };
constexpr int ACT_LAST = ACT_SWISH;
constexpr float LEAKY_SLOPE = 0.2f;

// Both helpers go straight to the hardware exp2 / log2 units (v_exp_f32 / v_log_f32, ~1 ulp; __expf/__logf expand to the
// denormal-safe sequences, ~3x the instructions, with divergent branches) plus a short series where the direct form would
// cancel, so that RELATIVE accuracy (a few 1e-6) holds down to saturated units with sigmoid ~ 1e-30: their gradients are
// tiny but RMSprop/Adam normalise them to full-size updates in the first steps, and a unit whose derivative underflowed to
// exactly 0 would silently stop moving (seen as a 13 % update mismatch on the tiny fixture when the series was dropped).
// FP32 MFMA and the vector ALU are one resource on gfx950 (scratch/mfma/samewave.hip), so every instruction here is paid
// for in matrix time: branch-free, ~11 instructions each.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// x < thr ? lo : hi as compare + v_cndmask.  Written as asm because the compiler turns a `?:` whose arms hold a transcendental
// into a divergent branch per element (saveexec / xor / or around each arm: +5 scalar instructions and no interleaving of
// neighbouring elements - 64 such diamonds per tile in the N-row epilogues).
__device__ __forceinline__ float select_lt(float x, float thr, float lo, float hi) {
  float r;
  asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(x), "v"(thr), "v"(hi), "v"(lo) : "vcc");
  return r;
}

__device__ __forceinline__ float softplus_f(float x) {
  // max(x,0) + log1p(exp(-|x|)) is the overflow-free form of log(1+exp(x)).  F.softplus' threshold (x > 20 -> x) needs no
  // select: there exp(-x) < 2.1e-9 is below half an ulp of x, so the sum rounds to x exactly.
  const float t = fast_exp(-fabsf(x));
  const float series = t * (1.f - t * (0.5f - t * (1.f / 3.f)));                  // log1p(t), |err| < t^4/4
  const float direct = __builtin_amdgcn_logf(1.f + t) * 0.693147180559945309f;    // rounding of 1+t: rel. err <= 6e-8 / t
  return fmaxf(x, 0.f) + select_lt(t, 8e-3f, series, direct);
}

// softplus'(pre) = sigmoid(pre) = 1 - exp(-a) from the saved output a = softplus(pre)
__device__ __forceinline__ float softplus_d1_from_out(float a) {
  const float series = a * (1.f - a * (0.5f - a * (1.f / 6.f)));
  return select_lt(a, 0.02f, series, 1.f - fast_exp(-a));
}

// Two elements at a time: everything that is not a transcendental or a select becomes one packed instruction
// (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two FP32 results per lane per issue slot).  Same arithmetic as the scalar forms.
__device__ __forceinline__ f32x2 softplus_f2(f32x2 x) {
  f32x2 t;
  t[0] = __builtin_amdgcn_exp2f(__builtin_fabsf(x[0]) * -1.44269504088896341f);
  t[1] = __builtin_amdgcn_exp2f(__builtin_fabsf(x[1]) * -1.44269504088896341f);
  const f32x2 series = t * (1.f - t * (0.5f - t * (1.f / 3.f)));
  const f32x2 w = 1.f + t;
  f32x2 lg;
  lg[0] = __builtin_amdgcn_logf(w[0]);
  lg[1] = __builtin_amdgcn_logf(w[1]);
  const f32x2 direct = lg * 0.693147180559945309f;
  f32x2 r;
  r[0] = fmaxf(x[0], 0.f);
  r[1] = fmaxf(x[1], 0.f);
  f32x2 l1p;
  l1p[0] = select_lt(t[0], 8e-3f, series[0], direct[0]);
  l1p[1] = select_lt(t[1], 8e-3f, series[1], direct[1]);
  return r + l1p;
}
// em = exp(-a) (= 1 - sigmoid) is handed back too: the CHAIN epilogue of the N-row kernel needs it
__device__ __forceinline__ f32x2 softplus_d1_from_out2(f32x2 a, f32x2& em) {
  const f32x2 t = a * -1.44269504088896341f;
  em[0] = __builtin_amdgcn_exp2f(t[0]);
  em[1] = __builtin_amdgcn_exp2f(t[1]);
  const f32x2 direct = 1.f - em;
  const f32x2 series = a * (1.f - a * (0.5f - a * (1.f / 6.f)));
  f32x2 r;
  r[0] = select_lt(a[0], 0.02f, series[0], direct[0]);
  r[1] = select_lt(a[1], 0.02f, series[1], direct[1]);
  return r;
}
// ---- swish (utils/models.py:8-10,29-30: x * sigmoid(x)) ---------------------------------------------------------------------------
// Every kernel rebuilds act' / act'' from the SAVED OUTPUT (one tensor per layer serves as the next layer's input and as the
// derivative's argument).  y = x sigmoid(x) is not monotonic: it falls from 0 to its minimum y_m = -0.27846 at x_m = -1.27846 and
// rises from there, so an output in (y_m, 0) has two pre-images.  The forward therefore records the BRANCH (x < x_m or not) in the
// lowest mantissa bit of the stored output - a perturbation of at most one ulp of y, below the rounding of the product that made it -
// and the derivative helpers recover x by 32 bisection steps on that branch (both branches are monotonic; bracket [-104, x_m] or
// [x_m, max(y, 0) + 0.2785], i.e. |x error| <= bracket / 2^32), then evaluate swish'(x) = s (1 + x (1 - s)) and
// swish''(x) = s (1 - s) (2 + x (1 - 2 s)) with s = sigmoid(x).  ~250 vector-ALU instructions per element: swish layers run on
// the generic kernels only (no shipped recipe uses it), where this is affordable.  Close to the minimum the inversion is
// ill-conditioned (dy/dx -> 0): there |swish' error| <= swish''(x_m) |x error| ~ 1e-4 at worst, on the few elements within 1e-3 of x_m.
constexpr float SWISH_XM = -1.2784645f;
__device__ __forceinline__ float sigmoid_f(float x) {
  const float t = fast_exp(-fabsf(x));             // in (0, 1]: no overflow
  const float r = 1.f / (1.f + t);
  return x >= 0.f ? r : t * r;
}
__device__ __forceinline__ float swish_raw(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float swish_f(float x) {
  const unsigned b = (__float_as_uint(swish_raw(x)) & ~1u) | (x < SWISH_XM ? 1u : 0u);
  return __uint_as_float(b);
}
__device__ __forceinline__ float swish_pre_from_out(float y) {
  const bool left = (__float_as_uint(y) & 1u) != 0u && y <= 0.f;
  float lo = left ? -104.f : SWISH_XM, hi = left ? SWISH_XM : fmaxf(y, 0.f) + 0.2785f;
#pragma nounroll
  for (int it = 0; it < 32; ++it) {
    const float mid = 0.5f * (lo + hi);
    const float fm = swish_raw(mid);
    const bool up = left ? (fm > y) : (fm < y);      // the root lies above mid (left branch: swish falls)
    lo = up ? mid : lo;
    hi = up ? hi : mid;
  }
  return 0.5f * (lo + hi);
}
__device__ __forceinline__ float swish_d1_from_out(float y) {
  const float x = swish_pre_from_out(y), s = sigmoid_f(x);
  return s * __builtin_fmaf(x, 1.f - s, 1.f);
}
__device__ __forceinline__ float swish_ratio_from_out(float y) {      // swish'' / swish'; 0 where swish' vanishes (the factor it multiplies carries swish')
  const float x = swish_pre_from_out(y), s = sigmoid_f(x);
  const float d1 = s * __builtin_fmaf(x, 1.f - s, 1.f);
  const float d2 = s * (1.f - s) * __builtin_fmaf(x, 1.f - 2.f * s, 2.f);
  return fabsf(d1) > 1e-20f ? d2 / d1 : 0.f;
}

template <int ACT>
__device__ __forceinline__ f32x2 act_fwd2(f32x2 x) {
  if (ACT == ACT_RELU) { f32x2 r; r[0] = fmaxf(x[0], 0.f); r[1] = fmaxf(x[1], 0.f); return r; }
  if (ACT == ACT_SOFTPLUS) return softplus_f2(x);
  return x;
}
template <int ACT>
__device__ __forceinline__ f32x2 act_d1_2(f32x2 a, f32x2& em) {
  if (ACT == ACT_RELU) { f32x2 r; r[0] = a[0] > 0.f ? 1.f : 0.f; r[1] = a[1] > 0.f ? 1.f : 0.f; em = 1.f - r; return r; }
  if (ACT == ACT_SOFTPLUS) return softplus_d1_from_out2(a, em);
  em = f32x2{0.f, 0.f};
  return f32x2{1.f, 1.f};
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if (ACT == ACT_SOFTPLUS) return softplus_f(x);
  if (ACT == ACT_ELU) return x > 0.f ? x : expm1f(x);
  if (ACT == ACT_TANH) return tanhf(x);
  if (ACT == ACT_LEAKY) return x > 0.f ? x : LEAKY_SLOPE * x;
  if (ACT == ACT_SWISH) return swish_f(x);
  return x;
}

// derivative s of the activation expressed through the saved post-activation value a
// (elu: a <= 0 means a = exp(x) - 1, s = exp(x) = a + 1; tanh: s = 1 - a^2; leaky: sign(a) = sign(x))
template <int ACT>
__device__ __forceinline__ float act_d1(float a) {
  if (ACT == ACT_RELU) return a > 0.f ? 1.f : 0.f;
  if (ACT == ACT_SOFTPLUS) return softplus_d1_from_out(a);
  if (ACT == ACT_ELU) return a > 0.f ? 1.f : a + 1.f;
  if (ACT == ACT_TANH) return __builtin_fmaf(-a, a, 1.f);
  if (ACT == ACT_LEAKY) return a > 0.f ? 1.f : LEAKY_SLOPE;
  if (ACT == ACT_SWISH) return swish_d1_from_out(a);
  return 1.f;
}

// second derivative over first, s' / s, from the saved output (the forward-mode pass through the score network multiplies a
// tensor that already carries s by it: EPI_CHAIN).  softplus: 1 - s = exp(-a) (no cancellation); elu: 1 on the exponential branch;
// tanh: -2a; piecewise linear activations: 0.
template <int ACT>
__device__ __forceinline__ float act_ratio(float a) {
  if (ACT == ACT_SOFTPLUS) return fast_exp(-a);
  if (ACT == ACT_ELU) return a > 0.f ? 0.f : 1.f;
  if (ACT == ACT_TANH) return -2.f * a;
  if (ACT == ACT_SWISH) return swish_ratio_from_out(a);
  return 0.f;
}

// ELU (F.elu, alpha = 1; utils/models.py:17-18, nn.ELU in models/ivae/resconv.py:23-31): x > 0 ? x : exp(x) - 1; its derivative
// from the saved OUTPUT a: a > 0 ? 1 : a + 1 (= exp(x))
__device__ __forceinline__ float act_fwd_rt(int act, float x) {
  switch (act) {
    case ACT_RELU: return act_fwd<ACT_RELU>(x);
    case ACT_SOFTPLUS: return act_fwd<ACT_SOFTPLUS>(x);
    case ACT_ELU: return act_fwd<ACT_ELU>(x);
    case ACT_TANH: return act_fwd<ACT_TANH>(x);
    case ACT_LEAKY: return act_fwd<ACT_LEAKY>(x);
    case ACT_SWISH: return act_fwd<ACT_SWISH>(x);
    default: return x;
  }
}
__device__ __forceinline__ float act_d1_rt(int act, float a) {
  switch (act) {
    case ACT_RELU: return act_d1<ACT_RELU>(a);
    case ACT_SOFTPLUS: return act_d1<ACT_SOFTPLUS>(a);
    case ACT_ELU: return act_d1<ACT_ELU>(a);
    case ACT_TANH: return act_d1<ACT_TANH>(a);
    case ACT_LEAKY: return act_d1<ACT_LEAKY>(a);
    case ACT_SWISH: return act_d1<ACT_SWISH>(a);
    default: return 1.f;
  }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ardae
