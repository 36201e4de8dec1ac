// odbox_common.h -- shared by the two translation units of libodbox.so.
#ifndef ODBOX_COMMON_H
#define ODBOX_COMMON_H
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "odbox.h"

namespace odbox {

// error plumbing (defined in odbox_train.hip)
int fail(int code, const char *fmt, ...);
const char *last_error();

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return odbox::fail(ODBOX_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ODBOX_EHIP, "launch %s: %s", what, hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
constexpr int WAVE = 64;
// ints between per-image atomic counters: returning atomics on one cache line serialise in L2
// (measured: 64 counters packed in 256 B cap a kernel at ~240 atomics/us chip-wide)
constexpr uint32_t CNT_STRIDE = 64;

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so give each
// XCD a contiguous run of tiles: neighbouring output tiles then meet in one L2.  Bijective for
// every grid size.  Affinity only, never correctness.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks)
{
    uint32_t xcd = bid & 7u, q = nblocks >> 3, r = nblocks & 7u;
    uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// v_min_f32 / v_max_f32 as the hardware has them, for the matcher's hot loops.  fminf / fmaxf are llvm.minnum /
// maxnum, which in IEEE mode put a canonicalising v_max_f32 x, x, x in front of every operand that comes out of a
// load (a signalling NaN must be quieted first): +4 instructions per (anchor, GT) pair.  -fno-honor-nans removed them
// (configs[3] 396 -> 343 us, the sweep's matcher waves 49.7 -> 39.1 us per launch, same-box A/B) but also told the
// compiler that NO operation of the unit ever sees a NaN, while the matcher's degenerate-box guards and the 0/0
// offsets of zero-size anchors rely on NaNs behaving as IEEE says.  So the unit is built WITHOUT that flag and only
// these two instructions are taken as they are: a number beats a quiet NaN, two numbers compare as in fminf / fmaxf.
// Plain asm (not volatile): the compiler may move, merge or drop them like any pure instruction.
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x / d and x % d for x*e < d*2^32 with m = floor(2^32 / d): one mulhi and one fix-up.
__device__ __forceinline__ void divmod_u32(uint32_t x, uint32_t d, uint32_t m, uint32_t &q, uint32_t &r)
{
    q = __umulhi(x, m);
    r = x - q * d;
    if (r >= d) { ++q; r -= d; }
}

// fp32 -> uint32 whose unsigned order is the float order (finite inputs; -0 folded into +0).
__device__ __forceinline__ uint32_t f32_ordered(float s)
{
    uint32_t u = __float_as_uint(s + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Candidate key  [class:12][~ordered(score):32][anchor:20] : ascending u64 order is
// (class asc, score desc, anchor asc) = the visiting order of per-class greedy NMS.
// Final key      [~ordered(score):32][class:12][anchor:20] : ascending order is (score desc,
// class asc, anchor asc) = tf.math.top_k over the class-ordered concat (bndbox.py:220-229):
// inside one class equal scores were selected in ascending anchor order.
__device__ __forceinline__ uint64_t cand_key(uint32_t cls, float score, uint32_t anchor)
{
    return ((uint64_t)cls << 52) | ((uint64_t)(~f32_ordered(score)) << 20) | anchor;
}
__device__ __forceinline__ uint32_t key_class(uint64_t k) { return (uint32_t)(k >> 52); }
__device__ __forceinline__ uint32_t key_anchor(uint64_t k) { return (uint32_t)k & 0xfffffu; }
__device__ __forceinline__ float key_score(uint64_t k) { return f32_unordered(~(uint32_t)(k >> 20)); }
__device__ __forceinline__ uint64_t cand_to_final(uint64_t k)
{
    return ((k >> 20) << 32) | ((k >> 52) << 20) | (k & 0xfffffu);
}
__device__ __forceinline__ uint64_t final_to_cand(uint64_t k)
{
    return (((k >> 20) & 0xfffu) << 52) | ((k >> 32) << 20) | (k & 0xfffffu);
}
// pre-NMS top-k order [~score:32][anchor:20][class:12]: (score desc, anchor asc)
__device__ __forceinline__ uint64_t cand_to_topk(uint64_t k)
{
    return (((k >> 20) & 0xffffffffull) << 32) | ((k & 0xfffffu) << 12) | (k >> 52);
}
__device__ __forceinline__ uint64_t topk_to_cand(uint64_t k)
{
    return ((k & 0xfffu) << 52) | ((k >> 32) << 20) | ((k >> 12) & 0xfffffu);
}

// ------------------------------------------------------------------------------------------
// elementwise box ops
// ------------------------------------------------------------------------------------------
// bndbox.py:88-99
__device__ __forceinline__ float4 regress_one(const float4 a, const float4 r)
{
    const float w = a.z - a.x, h = a.w - a.y;
    return make_float4(a.x + (r.x * 0.2f + 0.0f) * w, a.y + (r.y * 0.2f + 0.0f) * h,
                       a.z + (r.z * 0.2f + 0.0f) * w, a.w + (r.w * 0.2f + 0.0f) * h);
}
// bndbox.py:105-115
__device__ __forceinline__ float4 clip_one(const float4 v, float hi_x, float hi_y)
{
    return make_float4(fminf(fmaxf(v.x, 0.0f), hi_x), fminf(fmaxf(v.y, 0.0f), hi_y),
                       fminf(fmaxf(v.z, 0.0f), hi_x), fminf(fmaxf(v.w, 0.0f), hi_y));
}

}  // namespace odbox
#endif
