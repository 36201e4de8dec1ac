// odbox.hip -- hand-written gfx950 (MI355X, wave64) kernels + the C-ABI of include/odbox.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
// -ffp-contract=off is part of the contract: the reference evaluates every TensorFlow op as a
// separately rounded fp32 op, so no multiply-add here may be fused into an FMA; divisions are
// IEEE (hipcc's default correctly-rounded fp32 divide).  No MFMA: the path is elementwise and
// reduction work, HBM-bound except the dense-crowd matcher (VALU-bound).
//
// Reference citations are file:line under the reference tree (bioothod/object_detection).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "odbox.h"

namespace {

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(ODBOX_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
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

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so give each
// XCD a contiguous run of tiles: neighbouring output tiles then meet in one L2.  Bijective for
// every grid size.  Affinity only, never correctness.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks)
{
    uint32_t xcd = bid & 7u, q = nblocks >> 3, r = nblocks & 7u;
    uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
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
// anchors.py:60-87  tile base anchors over a feature map
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchor_tile_kernel(const float4 *__restrict__ base, int A, int fw,
                                                          float stride, float4 *__restrict__ out,
                                                          uint32_t total)
{
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    uint32_t cell = t / (uint32_t)A, a = t - cell * (uint32_t)A;
    uint32_t i = cell / (uint32_t)fw, j = cell - i * (uint32_t)fw;
    float sx = ((float)j + 0.5f) * stride;
    float sy = ((float)i + 0.5f) * stride;
    float4 b = base[a];
    out[t] = make_float4(b.x + sx, b.y + sy, b.z + sx, b.w + sy);
}

// ------------------------------------------------------------------------------------------
// bndbox.py:273-297  IoU of one (anchor, GT) pair, GT area precomputed
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float pair_iou(const float4 a, float a_area, const float4 g, float g_area)
{
    float ix0 = fmaxf(a.x, g.x);
    float ix1 = fminf(a.z, g.z);
    float iy0 = fmaxf(a.y, g.y);
    float iy1 = fminf(a.w, g.w);
    float i_area = fmaxf(ix1 - ix0, 0.0f) * fmaxf(iy1 - iy0, 0.0f);
    float u_area = ((a_area + g_area) - i_area) + 1e-8f;
    return i_area / u_area;
}

// ------------------------------------------------------------------------------------------
// Fused matcher + encoder: anchors.py:116-213 (+ :215-274, :277-299, bndbox.py:254-299).
//
// One block = one tile of ME_TILE anchors of one image.
//   phase 1  thread = anchor: GT boxes of the image staged in LDS (box, area, pad flag) in
//            chunks of ME_GCHUNK, IoU + strict-> argmax in registers (first max wins).
//   phase 2  thresholds, out-of-image test, label, offsets; compact outputs stored directly;
//            5 regression floats and a packed (label, state) word staged in LDS.
//   phase 3  the two dense outputs of the tile are contiguous in memory ([tile,5] and
//            [tile,C+1] floats): the block streams them as 16-byte stores over the FLAT element
//            range (head/tail handled with scalar stores), rebuilding each one-hot element from
//            the packed word.  This is the HBM-bound part: 4(C+1)+20 of the 4(C+1)+36 bytes.
// ------------------------------------------------------------------------------------------
constexpr int ME_TILE = 256;
constexpr int ME_GCHUNK = 512;

__global__ __launch_bounds__(ME_TILE) void match_encode_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ gt_boxes,
    const int32_t *__restrict__ gt_labels, int N, int G, int C, float img_h, float img_w,
    float neg_thr, float pos_thr, float *__restrict__ out_reg, float *__restrict__ out_lab,
    int32_t *__restrict__ out_idx, int8_t *__restrict__ out_state, float *__restrict__ out_iou,
    uint32_t tiles_per_image, uint32_t d_magic)
{
    __shared__ float4 s_gt[ME_GCHUNK];
    __shared__ float2 s_ga[ME_GCHUNK];          // (area, 1.0 if padded row else 0.0)
    __shared__ float s_reg[ME_TILE * 5];
    __shared__ uint32_t s_ls[ME_TILE];          // (label + 1) | (state + 1) << 16

    const uint32_t tid = threadIdx.x;
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t b = blk / tiles_per_image;
    const uint32_t n0 = (blk - b * tiles_per_image) * ME_TILE;
    const uint32_t n = n0 + tid;
    const bool live = n < (uint32_t)N;
    const uint32_t tile_n = min((uint32_t)ME_TILE, (uint32_t)N - n0);

    const float4 a = live ? anchors[n] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float a_area = (a.w - a.y) * (a.z - a.x);                       // bndbox.py:285

    // ---- phase 1: argmax_g IoU ------------------------------------------------------------
    int best = 0;
    float best_iou = -INFINITY;
    const float4 *gtb = gt_boxes + (size_t)b * G;
    for (int g0 = 0; g0 < G; g0 += ME_GCHUNK) {
        const int cnt = min(ME_GCHUNK, G - g0);
        if (g0) __syncthreads();
        for (int g = tid; g < cnt; g += ME_TILE) {
            float4 gb = gtb[g0 + g];
            float area = (gb.w - gb.y) * (gb.z - gb.x);                  // bndbox.py:286
            float gmax = fmaxf(fmaxf(gb.x, gb.y), fmaxf(gb.z, gb.w));    // bndbox.py:295
            s_gt[g] = gb;
            s_ga[g] = make_float2(area, gmax < 0.0f ? 1.0f : 0.0f);
        }
        __syncthreads();
#pragma unroll 4
        for (int g = 0; g < cnt; ++g) {
            float4 gb = s_gt[g];
            float2 ga = s_ga[g];
            float v = pair_iou(a, a_area, gb, ga.x);
            v = ga.y != 0.0f ? -1.0f : v;                                // bndbox.py:297
            if (v > best_iou) { best_iou = v; best = g0 + g; }           // tf.argmax: first max
        }
    }

    // ---- phase 2: states, label, offsets ---------------------------------------------------
    const bool positive = best_iou >= pos_thr;                            // anchors.py:267
    bool ignore = ((best_iou > neg_thr) && !positive) || (best_iou < 0.0f);   // :270-272
    const float xc = (a.x + a.z) / 2.0f;                                  // :178-184
    const float yc = (a.y + a.w) / 2.0f;
    ignore = ignore || (xc >= img_w) || (yc >= img_h);
    const int state = ignore ? -1 : (positive ? 1 : 0);
    int label = -1;
    if (live && positive && !ignore && gt_labels) label = gt_labels[(size_t)b * G + best];
    if (label < 0 || label >= C) label = -1;                              // tf.one_hot: zero row

    const size_t row = (size_t)b * N + n;
    if (live) {
        if (out_idx) out_idx[row] = best;
        if (out_state) out_state[row] = (int8_t)state;
        if (out_iou) out_iou[row] = best_iou;
    }
    if (out_reg) {                                                        // anchors.py:286-297
        const float4 g = gtb[best];
        const float aw = a.z - a.x, ah = a.w - a.y;
        s_reg[tid * 5 + 0] = (((g.x - a.x) / aw) - 0.0f) / 0.2f;
        s_reg[tid * 5 + 1] = (((g.y - a.y) / ah) - 0.0f) / 0.2f;
        s_reg[tid * 5 + 2] = (((g.z - a.z) / aw) - 0.0f) / 0.2f;
        s_reg[tid * 5 + 3] = (((g.w - a.w) / ah) - 0.0f) / 0.2f;
        s_reg[tid * 5 + 4] = (float)state;
    }
    s_ls[tid] = (uint32_t)(label + 1) | ((uint32_t)(state + 1) << 16);
    __syncthreads();

    // ---- phase 3a: [tile,5] regression rows, flat 16-byte stores ---------------------------
    if (out_reg) {
        float *p = out_reg + ((size_t)b * N + n0) * 5;
        const uint32_t count = tile_n * 5;
        const uint32_t head = min(count, (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) >> 2);
        const uint32_t nvec = (count - head) >> 2;
        if (tid < head) p[tid] = s_reg[tid];
        for (uint32_t v = tid; v < nvec; v += ME_TILE) {
            const uint32_t e = head + 4 * v;
            *reinterpret_cast<float4 *>(p + e) = make_float4(s_reg[e], s_reg[e + 1], s_reg[e + 2], s_reg[e + 3]);
        }
        const uint32_t done = head + 4 * nvec;
        if (tid < count - done) p[done + tid] = s_reg[done + tid];
    }

    // ---- phase 3b: [tile,C+1] one-hot + state rows, flat 16-byte stores --------------------
    if (out_lab) {
        const uint32_t d = (uint32_t)C + 1u;
        float *p = out_lab + ((size_t)b * N + n0) * d;
        const uint32_t count = tile_n * d;
        const uint32_t head = min(count, (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u) >> 2);
        const uint32_t nvec = (count - head) >> 2;
        auto element = [&](uint32_t ls, uint32_t col) -> float {
            const uint32_t lab1 = ls & 0xffffu;                      // label + 1, 0 = none
            const float st = (float)((int)(ls >> 16) - 1);
            return col == d - 1 ? st : (col + 1 == lab1 ? 1.0f : 0.0f);
        };
        if (tid < head) {
            uint32_t q, r;
            divmod_u32(tid, d, d_magic, q, r);
            p[tid] = element(s_ls[q], r);
        }
        for (uint32_t v = tid; v < nvec; v += ME_TILE) {
            const uint32_t e = head + 4 * v;
            uint32_t q, r;
            divmod_u32(e, d, d_magic, q, r);
            uint32_t ls = s_ls[q];
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = element(ls, r);
                if (++r == d) {
                    r = 0;
                    ++q;
                    ls = s_ls[min(q, (uint32_t)ME_TILE - 1)];
                }
            }
            *reinterpret_cast<float4 *>(p + e) = make_float4(o[0], o[1], o[2], o[3]);
        }
        const uint32_t done = head + 4 * nvec;
        if (tid < count - done) {
            uint32_t q, r;
            divmod_u32(done + tid, d, d_magic, q, r);
            p[done + tid] = element(s_ls[q], r);
        }
    }
}

// ------------------------------------------------------------------------------------------
// bndbox.py:254-299  dense [B,N,G] IoU (API parity; the training path never materialises it)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_overlap_kernel(const float4 *__restrict__ anchors,
                                                           const float4 *__restrict__ gt, int N, int G,
                                                           int anchors_batched, float *__restrict__ out,
                                                           size_t total)
{
    size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    size_t bn = t / (size_t)G;
    uint32_t g = (uint32_t)(t - bn * (size_t)G);
    size_t b = bn / (size_t)N;
    size_t n = bn - b * (size_t)N;
    float4 a = anchors[anchors_batched ? bn : n];
    float4 gb = gt[b * (size_t)G + g];
    float a_area = (a.w - a.y) * (a.z - a.x);
    float g_area = (gb.w - gb.y) * (gb.z - gb.x);
    float v = pair_iou(a, a_area, gb, g_area);
    float gmax = fmaxf(fmaxf(gb.x, gb.y), fmaxf(gb.z, gb.w));
    out[t] = gmax < 0.0f ? -1.0f : v;
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

__global__ __launch_bounds__(256) void regress_kernel(const float4 *__restrict__ boxes, int boxes_batched,
                                                      const float4 *__restrict__ reg, uint32_t N,
                                                      float4 *__restrict__ out, size_t total)
{
    size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    float4 a = boxes[boxes_batched ? t : t % N];
    out[t] = regress_one(a, reg[t]);
}

__global__ __launch_bounds__(256) void clip_kernel(const float4 *__restrict__ boxes, float hi_x, float hi_y,
                                                   float4 *__restrict__ out, size_t total)
{
    size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    out[t] = clip_one(boxes[t], hi_x, hi_y);
}

__global__ __launch_bounds__(256) void scale_kernel(const float4 *__restrict__ boxes, float fx, float fy,
                                                    int divide, float4 *__restrict__ out, size_t total)
{
    size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    float4 v = boxes[t];
    out[t] = divide ? make_float4(v.x / fx, v.y / fy, v.z / fx, v.w / fy)
                    : make_float4(v.x * fx, v.y * fy, v.z * fx, v.w * fy);
}

// anchors.py:286-297
__global__ __launch_bounds__(256) void transform_kernel(const float4 *__restrict__ anchors, int anchors_batched,
                                                        const float4 *__restrict__ gt, uint32_t N,
                                                        float4 *__restrict__ out, size_t total)
{
    size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    float4 a = anchors[anchors_batched ? t : t % N];
    float4 g = gt[t];
    const float aw = a.z - a.x, ah = a.w - a.y;
    out[t] = make_float4((((g.x - a.x) / aw) - 0.0f) / 0.2f, (((g.y - a.y) / ah) - 0.0f) / 0.2f,
                         (((g.z - a.z) / aw) - 0.0f) / 0.2f, (((g.w - a.w) / ah) - 0.0f) / 0.2f);
}

// ------------------------------------------------------------------------------------------
// decode + NMS, kernel 1: stream the [B,N,C] scores once (the HBM-bound kernel).
//
// One wave (= one 64-thread block) owns a tile of 64*RPL consecutive rows = one contiguous
// run of bytes.  The tile goes HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane when the
// run is 16-byte aligned, 4 B otherwise; no VGPR staging), then lane l reduces rows l, l+64..
// out of LDS.  Row r starts its scan at column (r * skew) mod C so that the 32 lanes of a
// ds_read_b32 group hit 32 different banks for every C (skew makes the lane stride odd); the
// rotated scan keeps tf.argmax's first-max rule through an explicit (value, column) compare.
// Rows with max score > threshold append one 64-bit key to the image's candidate list
// (one atomic per wave).  bndbox.py:153-154 (argmax / reduce_max), :205 (score filter).
// ------------------------------------------------------------------------------------------
template <int VEC>
__device__ __forceinline__ void dma_tile(const float *src, float *lds, uint32_t nfloat, uint32_t lane)
{
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    if constexpr (VEC == 4) {
        const uint32_t nvec = nfloat >> 2;
        uint32_t i = 0;
        for (; i + WAVE <= nvec; i += WAVE)
            __builtin_amdgcn_global_load_lds((gptr)(src + (size_t)(i + lane) * 4), (lptr)(lds + i * 4), 16, 0, 0);
        if (i < nvec) {   // last partial 16-byte row of lanes, then the <4 float tail
            if (i + lane < nvec)
                __builtin_amdgcn_global_load_lds((gptr)(src + (size_t)(i + lane) * 4), (lptr)(lds + i * 4), 16, 0, 0);
        }
        const uint32_t done = nvec << 2;
        if (done + lane < nfloat)
            __builtin_amdgcn_global_load_lds((gptr)(src + done + lane), (lptr)(lds + done), 4, 0, 0);
    } else {
        uint32_t i = 0;
        for (; i + WAVE <= nfloat; i += WAVE)
            __builtin_amdgcn_global_load_lds((gptr)(src + i + lane), (lptr)(lds + i), 4, 0, 0);
        if (i + lane < nfloat)
            __builtin_amdgcn_global_load_lds((gptr)(src + i + lane), (lptr)(lds + i), 4, 0, 0);
    }
}

__global__ __launch_bounds__(WAVE) void score_scan_kernel(
    const float *__restrict__ scores, int N, int C, int rpl, float score_thr,
    uint32_t tiles_per_image, int *__restrict__ cand_count, uint64_t *__restrict__ cand_keys,
    uint32_t cand_stride)
{
    extern __shared__ __attribute__((aligned(16))) float s_tile[];
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t b = blk / tiles_per_image;
    const uint32_t rows_per_tile = (uint32_t)rpl * WAVE;
    const uint32_t n0 = (blk - b * tiles_per_image) * rows_per_tile;
    const uint32_t rows = min(rows_per_tile, (uint32_t)N - n0);
    const float *src = scores + ((size_t)b * N + n0) * (size_t)C;
    const uint32_t nfloat = rows * (uint32_t)C;

    if (((uintptr_t)src & 15u) == 0)
        dma_tile<4>(src, s_tile, nfloat, lane);
    else
        dma_tile<1>(src, s_tile, nfloat, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    const uint32_t skew = (C & 1) ? 0u : 1u;       // lane stride C + skew is odd
    for (int j = 0; j < rpl; ++j) {
        const uint32_t r = (uint32_t)j * WAVE + lane;
        const bool live = r < rows;
        float best = -INFINITY;
        uint32_t bcol = 0;
        if (live) {
            const float *rowp = s_tile + r * (uint32_t)C;
            uint32_t col = (r * skew) % (uint32_t)C;
            for (int k = 0; k < C; ++k) {
                const float v = rowp[col];
                const bool better = (v > best) || (v == best && col < bcol);
                best = better ? v : best;
                bcol = better ? col : bcol;
                col = (col + 1 == (uint32_t)C) ? 0u : col + 1;
            }
        }
        const bool cand = live && (best > score_thr);                     // bndbox.py:205
        const uint64_t mask = __ballot(cand);
        if (mask) {
            int base = 0;
            if (lane == 0) base = atomicAdd(cand_count + b, (int)__popcll(mask));
            base = __shfl(base, 0);
            if (cand) {
                const uint32_t pos = (uint32_t)base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                if (pos < cand_stride)
                    cand_keys[(size_t)b * cand_stride + pos] = cand_key(bcol, best, n0 + r);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Block-wide bitonic sort of P (power of two) u64 keys, ascending, in LDS or global memory.
// ------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 1024;
constexpr uint32_t SORT_LDS_CAP = 16384;   // keys; 128 KiB of LDS

__device__ void bitonic_sort(uint64_t *keys, uint32_t P)
{
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (P >> 1); t += SORT_THREADS) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                const uint64_t x = keys[lo], y = keys[hi];
                const bool asc = (lo & k) == 0;
                if ((x > y) == asc) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// kernel 2: per image, sort the candidate keys (class asc, score desc, anchor asc) and record
// where each class starts.  Optional pre-NMS top-k (extension, see odbox.h).
__global__ __launch_bounds__(SORT_THREADS) void cand_sort_kernel(
    int *__restrict__ cand_count, uint64_t *__restrict__ cand_keys, uint32_t cand_stride, int C,
    int pre_nms_topk, int *__restrict__ seg_start)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
    const uint32_t b = blockIdx.x;
    uint64_t *gk = cand_keys + (size_t)b * cand_stride;
    uint32_t M = min((uint32_t)cand_count[b], cand_stride);
    uint32_t P = max(next_pow2(M), 2u);
    const bool in_lds = P <= SORT_LDS_CAP;
    uint64_t *k = in_lds ? s_keys : gk;

    const bool topk = pre_nms_topk > 0 && M > (uint32_t)pre_nms_topk;
    for (uint32_t i = threadIdx.x; i < P; i += SORT_THREADS) {
        uint64_t v = i < M ? gk[i] : ~0ull;
        if (topk && i < M) v = cand_to_topk(v);
        k[i] = v;
    }
    __syncthreads();
    if (topk) {
        bitonic_sort(k, P);
        const uint32_t keep = (uint32_t)pre_nms_topk;
        for (uint32_t i = threadIdx.x; i < P; i += SORT_THREADS) k[i] = i < keep ? topk_to_cand(k[i]) : ~0ull;
        M = keep;
        __syncthreads();
    }
    bitonic_sort(k, P);

    int *seg = seg_start + (size_t)b * (C + 1);
    for (uint32_t i = threadIdx.x; i <= M; i += SORT_THREADS) {
        if (i < M && in_lds) gk[i] = k[i];
        const int c_hi = i < M ? (int)key_class(k[i]) : C;
        const int c_lo = i > 0 ? (int)key_class(k[i - 1]) + 1 : 0;
        for (int c = c_lo; c <= min(c_hi, C); ++c) seg[c] = (int)i;
    }
    if (threadIdx.x == 0) cand_count[b] = (int)M;
}

// decoded (+ clipped) box of anchor `idx` of image b: bndbox.py:88-99, :105-115
__device__ __forceinline__ float4 candidate_box(const float4 *__restrict__ anchors,
                                                const float4 *__restrict__ rb, size_t b, int N,
                                                uint32_t idx, int do_clip, float hi_x, float hi_y)
{
    float4 v = rb[b * (size_t)N + idx];
    if (anchors) v = regress_one(anchors[idx], v);
    if (do_clip) v = clip_one(v, hi_x, hi_y);
    return v;
}

// TF NonMaxSuppressionV3 IoU on min/max-normalised boxes with precomputed areas
// (third-party kernel semantics; call site bndbox.py:170-175)
__device__ __forceinline__ float nms_iou(const float4 p, float pa, const float4 q, float qa)
{
    if (pa <= 0.0f || qa <= 0.0f) return 0.0f;
    const float ih = fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y), 0.0f);
    const float iw = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x), 0.0f);
    const float inter = ih * iw;
    return inter / ((pa + qa) - inter);
}

// kernel 3: greedy NMS, one wave per (image, class) segment of the sorted candidate list.
// The kept list (<= max_ret boxes) lives in LDS; candidates are taken 64 at a time: first
// tested against the kept list, then resolved inside the chunk with ballots (the lowest live
// lane is kept and suppresses the lanes above it).  Stops at max_ret kept (bndbox.py:172).
constexpr int NMS_WAVES = 4;

__global__ __launch_bounds__(NMS_WAVES *WAVE) void nms_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ rb, int N, int C, int do_clip,
    float hi_x, float hi_y, float iou_thr, int max_ret, const uint64_t *__restrict__ cand_keys,
    uint32_t cand_stride, const int *__restrict__ seg_start, int *__restrict__ fin_count,
    uint64_t *__restrict__ fin_keys, uint32_t fin_stride, uint32_t total_segments, int waves_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg_id = blockIdx.x * (uint32_t)waves_per_block + wave;
    if (seg_id >= total_segments) return;
    const uint32_t b = seg_id / (uint32_t)C;
    const uint32_t c = seg_id - b * (uint32_t)C;
    const int start = seg_start[(size_t)b * (C + 1) + c];
    const int end = seg_start[(size_t)b * (C + 1) + c + 1];
    if (end <= start) return;

    // per-wave LDS: max_ret boxes (float4) + max_ret areas + max_ret keys
    const size_t per_wave = (size_t)max_ret * (16 + 4 + 8);
    unsigned char *mine = s_raw + wave * ((per_wave + 15) & ~(size_t)15);
    float4 *s_box = reinterpret_cast<float4 *>(mine);
    uint64_t *s_key = reinterpret_cast<uint64_t *>(mine + (size_t)max_ret * 16);
    float *s_area = reinterpret_cast<float *>(mine + (size_t)max_ret * 24);

    const uint64_t *keys = cand_keys + (size_t)b * cand_stride;
    int kept = 0;
    for (int base = start; base < end && kept < max_ret; base += WAVE) {
        const int i = base + (int)lane;
        const bool valid = i < end;
        const uint64_t key = valid ? keys[i] : 0ull;
        float4 nb = make_float4(0.f, 0.f, 0.f, 0.f);
        float area = 0.0f;
        if (valid) {
            const float4 v = candidate_box(anchors, rb, b, N, key_anchor(key), do_clip, hi_x, hi_y);
            nb = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
            area = (nb.w - nb.y) * (nb.z - nb.x);
        }
        bool alive = valid;
        for (int k = 0; k < kept; ++k) {
            if (alive && nms_iou(nb, area, s_box[k], s_area[k]) > iou_thr) alive = false;
            if (__ballot(alive) == 0ull) break;
        }
        uint64_t mask = __ballot(alive);
        while (mask != 0ull && kept < max_ret) {
            const int j = __ffsll((unsigned long long)mask) - 1;
            const float4 jb = make_float4(__shfl(nb.x, j), __shfl(nb.y, j), __shfl(nb.z, j), __shfl(nb.w, j));
            const float ja = __shfl(area, j);
            if ((int)lane == j) {
                s_box[kept] = nb;
                s_area[kept] = area;
                s_key[kept] = key;
            }
            ++kept;
            if (alive && (int)lane > j && nms_iou(nb, area, jb, ja) > iou_thr) alive = false;
            mask = __ballot(alive) & ~((2ull << j) - 1ull);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // append this segment's kept keys to the image's final list (order fixed by the next sort)
    int out_base = 0;
    if (lane == 0) out_base = atomicAdd(fin_count + b, kept);
    out_base = __shfl(out_base, 0);
    for (int k = (int)lane; k < kept; k += WAVE) {
        const uint32_t pos = (uint32_t)(out_base + k);
        if (pos < fin_stride) fin_keys[(size_t)b * fin_stride + pos] = cand_to_final(s_key[k]);
    }
}

// kernel 4: per image, order the kept detections like tf.math.top_k over the class-ordered
// concat (bndbox.py:220-229), emit the first max_ret and zero-pad the rest (:233-244).
__global__ __launch_bounds__(SORT_THREADS) void finalize_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ rb, int N, int do_clip, float hi_x,
    float hi_y, int max_ret, const int *__restrict__ fin_count, uint64_t *__restrict__ fin_keys,
    uint32_t fin_stride, float4 *__restrict__ out_boxes, float *__restrict__ out_scores,
    int32_t *__restrict__ out_labels, int32_t *__restrict__ out_keep, int32_t *__restrict__ out_count)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
    const uint32_t b = blockIdx.x;
    uint64_t *gk = fin_keys + (size_t)b * fin_stride;
    const uint32_t n = min((uint32_t)fin_count[b], fin_stride);
    const uint32_t P = max(next_pow2(n), 2u);
    const bool in_lds = P <= SORT_LDS_CAP;
    uint64_t *k = in_lds ? s_keys : gk;
    for (uint32_t i = threadIdx.x; i < P; i += SORT_THREADS) k[i] = i < n ? gk[i] : ~0ull;
    __syncthreads();
    if (n > 1) bitonic_sort(k, P);

    const uint32_t keep = min(n, (uint32_t)max_ret);
    for (uint32_t r = threadIdx.x; r < (uint32_t)max_ret; r += SORT_THREADS) {
        const size_t o = (size_t)b * max_ret + r;
        if (r < keep) {
            const uint64_t ck = final_to_cand(k[r]);
            const uint32_t idx = key_anchor(ck);
            out_boxes[o] = candidate_box(anchors, rb, b, N, idx, do_clip, hi_x, hi_y);
            out_scores[o] = key_score(ck);
            out_labels[o] = (int32_t)key_class(ck);
            if (out_keep) out_keep[o] = (int32_t)idx;
        } else {
            out_boxes[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[o] = 0.0f;
            out_labels[o] = 0;
            if (out_keep) out_keep[o] = -1;
        }
    }
    if (out_count && threadIdx.x == 0) out_count[b] = (int32_t)keep;
}

// ------------------------------------------------------------------------------------------
// workspace layout of odbox_decode_nms
// ------------------------------------------------------------------------------------------
struct NmsLayout {
    size_t off_counts, off_seg, off_cand, off_fin, total;
    uint32_t cand_stride, fin_stride;
    int rpl;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t host_next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

bool nms_layout(int B, int N, int C, int max_ret, NmsLayout &L)
{
    if (B < 1 || N < 1 || C < 1 || max_ret < 1) return false;
    if (N > ODBOX_MAX_ANCHORS || C > ODBOX_MAX_CLASSES || max_ret > ODBOX_MAX_RET) return false;
    // strides are powers of two whenever the global-memory sort could be taken
    L.cand_stride = (uint32_t)N > SORT_LDS_CAP ? host_next_pow2((uint32_t)N) : (uint32_t)(N < 2 ? 2 : N);
    uint64_t fin = (uint64_t)C * (uint64_t)max_ret;
    if (fin > (uint64_t)N) fin = (uint64_t)N;
    L.fin_stride = fin > SORT_LDS_CAP ? host_next_pow2((uint32_t)fin) : (uint32_t)(fin < 2 ? 2 : fin);
    L.off_counts = 0;
    L.off_seg = align_up((size_t)2 * B * sizeof(int), 256);
    L.off_cand = L.off_seg + align_up((size_t)B * (C + 1) * sizeof(int), 256);
    L.off_fin = L.off_cand + align_up((size_t)B * L.cand_stride * sizeof(uint64_t), 256);
    L.total = L.off_fin + align_up((size_t)B * L.fin_stride * sizeof(uint64_t), 256);
    // rows per lane of the score scan: aim at ~16-24 KiB of LDS per wave
    int rpl = 96 / C;
    L.rpl = rpl < 1 ? 1 : (rpl > 8 ? 8 : rpl);
    return true;
}

constexpr int SCAN_MAX_CLASSES_LDS = 512;   // 64 rows * 512 classes * 4 B = 128 KiB

// dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel
template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return 0;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" {

int odbox_version(void) { return ODBOX_VERSION; }
const char *odbox_last_error(void) { return g_err; }

int odbox_anchor_tile(const float *base, int A, int fh, int fw, float stride, float *out, void *stream)
{
    if (A < 1 || fh < 0 || fw < 0) return fail(ODBOX_EINVAL, "odbox_anchor_tile: bad shape A=%d fh=%d fw=%d", A, fh, fw);
    if (fh == 0 || fw == 0) return 0;
    if (!base || !out) return fail(ODBOX_EINVAL, "odbox_anchor_tile: NULL pointer");
    if (!aligned16(base) || !aligned16(out)) return fail(ODBOX_EINVAL, "odbox_anchor_tile: pointers must be 16-byte aligned");
    const uint64_t total = (uint64_t)fh * fw * A;
    if (total == 0) return 0;
    if (total > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_anchor_tile: too many anchors");
    hipLaunchKernelGGL(anchor_tile_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)base, A, fw, stride, (float4 *)out, (uint32_t)total);
    return launch_status("anchor_tile");
}

int odbox_match_encode(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N,
                       int G, int C, float img_h, float img_w, float neg_thr, float pos_thr,
                       float *out_regression, float *out_labels, int32_t *out_match_idx, int8_t *out_state,
                       float *out_max_iou, void *stream)
{
    if (B < 0 || N < 0 || G < 1 || C < 1 || C > ODBOX_MAX_CLASSES)
        return fail(ODBOX_EINVAL, "odbox_match_encode: bad shape B=%d N=%d G=%d C=%d (need G>=1, 1<=C<=%d)", B, N, G, C,
                    ODBOX_MAX_CLASSES);
    if (B == 0 || N == 0) return 0;
    if (!anchors || !gt_boxes) return fail(ODBOX_EINVAL, "odbox_match_encode: NULL input");
    if (out_labels && !gt_labels) return fail(ODBOX_EINVAL, "odbox_match_encode: out_labels needs gt_labels");
    if (!aligned16(anchors) || !aligned16(gt_boxes))
        return fail(ODBOX_EINVAL, "odbox_match_encode: anchors / gt_boxes must be 16-byte aligned");
    if (B == 0 || N == 0) return 0;
    const uint64_t tiles = ((uint64_t)N + ME_TILE - 1) / ME_TILE;
    const uint64_t blocks = tiles * (uint64_t)B;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_match_encode: grid too large");
    const uint32_t d = (uint32_t)C + 1u;
    const uint32_t magic = d == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / d);
    hipLaunchKernelGGL(match_encode_kernel, dim3((uint32_t)blocks), dim3(ME_TILE), 0, (hipStream_t)stream,
                       (const float4 *)anchors, (const float4 *)gt_boxes, gt_labels, N, G, C, img_h, img_w,
                       neg_thr, pos_thr, out_regression, out_labels, out_match_idx, out_state, out_max_iou,
                       (uint32_t)tiles, magic);
    return launch_status("match_encode");
}

int odbox_bbox_overlap(const float *anchors, const float *gt_boxes, int B, int N, int G, int anchors_batched,
                       float *out_iou, void *stream)
{
    if (B < 0 || N < 0 || G < 0) return fail(ODBOX_EINVAL, "odbox_bbox_overlap: bad shape");
    if ((size_t)B * N * G == 0) return 0;
    if (!anchors || !gt_boxes || !out_iou) return fail(ODBOX_EINVAL, "odbox_bbox_overlap: NULL pointer");
    if (!aligned16(anchors) || !aligned16(gt_boxes))
        return fail(ODBOX_EINVAL, "odbox_bbox_overlap: inputs must be 16-byte aligned");
    const size_t total = (size_t)B * N * G;
    if (total == 0) return 0;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_bbox_overlap: grid too large");
    hipLaunchKernelGGL(bbox_overlap_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)anchors, (const float4 *)gt_boxes, N, G, anchors_batched, out_iou, total);
    return launch_status("bbox_overlap");
}

static int elementwise_check(const char *name, const void *a, const void *b, const void *c, size_t total,
                             size_t &blocks)
{
    blocks = 0;
    if (total == 0) return 0;
    if (!a || !b || !c) return fail(ODBOX_EINVAL, "%s: NULL pointer", name);
    if (!aligned16(a) || !aligned16(b) || !aligned16(c)) return fail(ODBOX_EINVAL, "%s: pointers must be 16-byte aligned", name);
    blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "%s: grid too large", name);
    return 0;
}

int odbox_bbox_transform(const float *anchors, int anchors_batched, const float *gt_boxes, int B, int N,
                         float *out, void *stream)
{
    if (B < 0 || N < 0) return fail(ODBOX_EINVAL, "odbox_bbox_transform: bad shape");
    const size_t total = (size_t)B * N;
    size_t blocks;
    if (int rc = elementwise_check("odbox_bbox_transform", anchors, gt_boxes, out, total, blocks)) return rc;
    if (total == 0) return 0;
    hipLaunchKernelGGL(transform_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)anchors, anchors_batched, (const float4 *)gt_boxes, (uint32_t)N,
                       (float4 *)out, total);
    return launch_status("bbox_transform");
}

int odbox_regress_boxes(const float *boxes, int boxes_batched, const float *regressors, int B, int N,
                        float *out, void *stream)
{
    if (B < 0 || N < 0) return fail(ODBOX_EINVAL, "odbox_regress_boxes: bad shape");
    const size_t total = (size_t)B * N;
    size_t blocks;
    if (int rc = elementwise_check("odbox_regress_boxes", boxes, regressors, out, total, blocks)) return rc;
    if (total == 0) return 0;
    hipLaunchKernelGGL(regress_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes, boxes_batched, (const float4 *)regressors, (uint32_t)N,
                       (float4 *)out, total);
    return launch_status("regress_boxes");
}

int odbox_clip_boxes(const float *boxes, size_t n, float img_h, float img_w, float *out, void *stream)
{
    size_t blocks;
    if (int rc = elementwise_check("odbox_clip_boxes", boxes, boxes, out, n, blocks)) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(clip_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes, img_w - 1.0f, img_h - 1.0f, (float4 *)out, n);
    return launch_status("clip_boxes");
}

int odbox_scale_boxes(const float *boxes, size_t n, float fx, float fy, int divide, float *out, void *stream)
{
    size_t blocks;
    if (int rc = elementwise_check("odbox_scale_boxes", boxes, boxes, out, n, blocks)) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes, fx, fy, divide, (float4 *)out, n);
    return launch_status("scale_boxes");
}

size_t odbox_decode_nms_workspace(int B, int N, int C, int max_ret, int pre_nms_topk)
{
    (void)pre_nms_topk;
    NmsLayout L;
    if (!nms_layout(B, N, C, max_ret, L)) return 0;
    return L.total;
}

int odbox_decode_nms(const float *anchors, const float *regress_or_boxes, const float *scores, int B, int N,
                     int C, float img_h, float img_w, int do_clip, float score_thr, float iou_thr, int max_ret,
                     int pre_nms_topk, float *out_boxes, float *out_scores, int32_t *out_labels,
                     int32_t *out_keep_anchor, int32_t *out_count, void *workspace, size_t workspace_bytes,
                     void *stream)
{
    if (!regress_or_boxes || !scores || !out_boxes || !out_scores || !out_labels)
        return fail(ODBOX_EINVAL, "odbox_decode_nms: NULL pointer");
    if (B == 0) return 0;
    NmsLayout L;
    if (!nms_layout(B, N, C, max_ret, L))
        return fail(ODBOX_EINVAL,
                    "odbox_decode_nms: bad shape B=%d N=%d C=%d max_ret=%d (limits: N<=%d C<=%d max_ret<=%d)", B, N, C,
                    max_ret, ODBOX_MAX_ANCHORS, ODBOX_MAX_CLASSES, ODBOX_MAX_RET);
    if (C > SCAN_MAX_CLASSES_LDS)
        return fail(ODBOX_EINVAL, "odbox_decode_nms: C=%d exceeds the score-scan tile limit %d", C, SCAN_MAX_CLASSES_LDS);
    if (pre_nms_topk < 0) return fail(ODBOX_EINVAL, "odbox_decode_nms: pre_nms_topk < 0");
    if (!workspace || workspace_bytes < L.total)
        return fail(ODBOX_ENOSPC, "odbox_decode_nms: workspace %zu bytes, need %zu", workspace_bytes, L.total);
    if (((uintptr_t)workspace & 255u) != 0) return fail(ODBOX_EINVAL, "odbox_decode_nms: workspace must be 256-byte aligned");
    if ((anchors && !aligned16(anchors)) || !aligned16(regress_or_boxes) || !aligned16(out_boxes) ||
        ((uintptr_t)scores & 3u))
        return fail(ODBOX_EINVAL, "odbox_decode_nms: boxes must be 16-byte aligned, scores 4-byte aligned");

    hipStream_t st = (hipStream_t)stream;
    unsigned char *ws = (unsigned char *)workspace;
    int *cand_count = (int *)(ws + L.off_counts);
    int *fin_count = cand_count + B;
    int *seg_start = (int *)(ws + L.off_seg);
    uint64_t *cand_keys = (uint64_t *)(ws + L.off_cand);
    uint64_t *fin_keys = (uint64_t *)(ws + L.off_fin);
    const float hi_x = img_w - 1.0f, hi_y = img_h - 1.0f;

    HIP_TRY(hipMemsetAsync(cand_count, 0, (size_t)2 * B * sizeof(int), st));

    // kernel 1: score scan
    const uint32_t rows_per_tile = (uint32_t)L.rpl * WAVE;
    const uint64_t tiles = ((uint64_t)N + rows_per_tile - 1) / rows_per_tile;
    if (tiles * B > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_decode_nms: grid too large");
    const size_t scan_lds = (size_t)rows_per_tile * C * sizeof(float);
    if (int rc = allow_lds(score_scan_kernel, scan_lds)) return rc;
    hipLaunchKernelGGL(score_scan_kernel, dim3((uint32_t)(tiles * B)), dim3(WAVE), scan_lds, st, scores, N, C, L.rpl,
                       score_thr, (uint32_t)tiles, cand_count, cand_keys, L.cand_stride);
    if (int rc = launch_status("score_scan")) return rc;

    // kernel 2: per-image candidate sort + class segment starts
    const size_t sort_lds = (size_t)(L.cand_stride < SORT_LDS_CAP ? host_next_pow2(L.cand_stride) : SORT_LDS_CAP) * 8;
    if (int rc = allow_lds(cand_sort_kernel, sort_lds)) return rc;
    hipLaunchKernelGGL(cand_sort_kernel, dim3(B), dim3(SORT_THREADS), sort_lds, st, cand_count, cand_keys,
                       L.cand_stride, C, pre_nms_topk, seg_start);
    if (int rc = launch_status("cand_sort")) return rc;

    // kernel 3: per-(image, class) greedy NMS
    const size_t per_wave = align_up((size_t)max_ret * 28, 16);
    const int waves = per_wave * NMS_WAVES <= 64 * 1024 ? NMS_WAVES : 1;
    const uint64_t segs = (uint64_t)B * C;
    if (int rc = allow_lds(nms_kernel, per_wave * waves)) return rc;
    hipLaunchKernelGGL(nms_kernel, dim3((uint32_t)((segs + waves - 1) / waves)), dim3(waves * WAVE), per_wave * waves,
                       st, (const float4 *)anchors, (const float4 *)regress_or_boxes, N, C, do_clip, hi_x, hi_y,
                       iou_thr, max_ret, cand_keys, L.cand_stride, seg_start, fin_count, fin_keys, L.fin_stride,
                       (uint32_t)segs, waves);
    if (int rc = launch_status("nms")) return rc;

    // kernel 4: final order, top max_ret, zero padding
    const size_t fin_lds = (size_t)(L.fin_stride < SORT_LDS_CAP ? host_next_pow2(L.fin_stride) : SORT_LDS_CAP) * 8;
    if (int rc = allow_lds(finalize_kernel, fin_lds)) return rc;
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(SORT_THREADS), fin_lds, st, (const float4 *)anchors,
                       (const float4 *)regress_or_boxes, N, do_clip, hi_x, hi_y, max_ret, fin_count, fin_keys,
                       L.fin_stride, (float4 *)out_boxes, out_scores, out_labels, out_keep_anchor, out_count);
    return launch_status("finalize");
}

}  // extern "C"
