// odbox_eval.hip -- evaluation side of the box path: the step right after the
// all-gather of detections.  evaluate.py:14-55,93-110,172-181 (tensors -> COCO xywh records and
// areas), cocoeval.py:167-194 (computeIoU through pycocotools' bbIou) and cocoeval.py:239-317
// (evaluateImg: greedy matching per image, category, area range and IoU threshold), on device
// tensors instead of Python lists of dicts.
//
// Numerical contract: bbIou works on C doubles made from fp32 values (the reference hands
// `.numpy().tolist()` lists to pycocotools), so everything here is IEEE fp64 with the same
// association and no contraction (-ffp-contract=off): match tables are bit-exact, IoUs too.
#include "odbox_common.h"
#include <hip/hip_fp16.h>

using namespace odbox;

namespace {

// pycocotools maskApi.c bbIou for one (detection, ground truth) pair, iscrowd = 0 (evaluate.py:52
// never sets it).  Boxes are (x, y, w, h) with w, h the fp32 differences of evaluate.py:19-20.
__device__ __forceinline__ double bb_iou(float dx, float dy, float dw, float dh, float gx, float gy, float gw,
                                         float gh)
{
    const double Dx = dx, Dy = dy, Dw = dw, Dh = dh, Gx = gx, Gy = gy, Gw = gw, Gh = gh;
    const double w = fmin(Dw + Dx, Gw + Gx) - fmax(Dx, Gx);
    if (w <= 0.0) return 0.0;
    const double h = fmin(Dh + Dy, Gh + Gy) - fmax(Dy, Gy);
    if (h <= 0.0) return 0.0;
    const double i = w * h;
    const double u = (Dw * Dh + Gw * Gh) - i;
    return i / u;
}

// evaluate.py:19-22 / :39-45: xyxy -> (x, y, w, h); :41 fp32 area of a GT box; :177 double area of
// a detection (product of the two widened fp32 sides, exact).
__global__ __launch_bounds__(256) void coco_records_kernel(const float4 *__restrict__ boxes, size_t n,
                                                           float4 *__restrict__ xywh, float *__restrict__ area32,
                                                           double *__restrict__ area64)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 b = boxes[i];
    const float h = b.w - b.y, w = b.z - b.x;
    if (xywh) xywh[i] = make_float4(b.x, b.y, w, h);
    if (area32) area32[i] = h * w;
    if (area64) area64[i] = (double)w * (double)h;
}

// cocoeval.py:167-194 without the per-category regrouping: out[b, d, g] for every detection slot d
// and GT row g of image b (the caller slices by category / score order; padded rows give 0).
__global__ __launch_bounds__(256) void coco_iou_kernel(const float4 *__restrict__ dt, const float4 *__restrict__ gt,
                                                       int R, int G, double *__restrict__ out, size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const size_t bd = i / (size_t)G;
    const uint32_t g = (uint32_t)(i - bd * (size_t)G);
    const size_t b = bd / (size_t)R;
    const float4 D = dt[bd], Gb = gt[b * (size_t)G + g];
    out[i] = bb_iou(D.x, D.y, D.z - D.x, D.w - D.y, Gb.x, Gb.y, Gb.z - Gb.x, Gb.w - Gb.y);
}

// Per image: the evaluation order of its detections.  cocoeval.py:100-101 keeps a detection only if
// its category is known; :177-180 / :263-264 sort every (image, category) list by score with a
// stable mergesort (ties keep slot order) and cut it at maxDets[-1].  Categories never interact, so
// one list per image in (score desc, slot asc) order visits every category's list in its own order.
//   order[b, 0..count[b])  slots of the evaluated detections in that order
//   rank[b, slot]          position inside its (image, category) list, -1 = not evaluated
constexpr int ORDER_THREADS = 256;
__global__ __launch_bounds__(ORDER_THREADS) void coco_order_kernel(const float *__restrict__ scores,
                                                                   const int32_t *__restrict__ labels, int R, int C,
                                                                   int max_det, int32_t *__restrict__ order,
                                                                   int32_t *__restrict__ count,
                                                                   int32_t *__restrict__ rank)
{
    extern __shared__ unsigned char smem[];
    float *s_score = reinterpret_cast<float *>(smem);
    int32_t *s_label = reinterpret_cast<int32_t *>(smem) + R;          // label, or -1: unknown category
    uint8_t *s_keep = smem + (size_t)R * 8;                             // 1: evaluated
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const float *sc = scores + (size_t)b * R;
    const int32_t *lb = labels + (size_t)b * R;
    for (int i = tid; i < R; i += ORDER_THREADS) {
        const int32_t l = lb[i];
        s_score[i] = sc[i];
        s_label[i] = (l >= 0 && l < C) ? l : -1;
    }
    __syncthreads();
    // pass 1: rank inside the category (every detection of the category counts, kept or not)
    for (int i = tid; i < R; i += ORDER_THREADS) {
        int r = -1;
        const int32_t l = s_label[i];
        if (l >= 0) {
            const float s = s_score[i];
            r = 0;
            for (int j = 0; j < R; ++j) {
                const float sj = s_score[j];
                r += (s_label[j] == l && (sj > s || (sj == s && j < i))) ? 1 : 0;
            }
            if (r >= max_det) r = -1;
        }
        rank[(size_t)b * R + i] = r;
        s_keep[i] = r >= 0 ? 1 : 0;
    }
    __syncthreads();
    // pass 2: position among the evaluated detections of the image
    for (int i = tid; i < R; i += ORDER_THREADS) {
        if (!s_keep[i]) continue;
        const float s = s_score[i];
        int pos = 0;
        for (int j = 0; j < R; ++j) {
            const float sj = s_score[j];
            pos += (s_keep[j] && (sj > s || (sj == s && j < i))) ? 1 : 0;
        }
        order[(size_t)b * R + pos] = i;
    }
    if (tid == 0) {
        int n_eval = 0;
        for (int j = 0; j < R; ++j) n_eval += s_keep[j];
        count[b] = n_eval;
    }
}

constexpr int COCO_MAX_T = ODBOX_COCO_MAX_THRS;
constexpr int COCO_MAX_A = ODBOX_COCO_MAX_AREAS;
struct CocoParams {
    int T, A, C;
    double thr[COCO_MAX_T];
    double lo[COCO_MAX_A], hi[COCO_MAX_A];
};

__device__ __forceinline__ float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// cocoeval.py:239-317 evaluateImg.  One wave per (image, area range, IoU threshold); lane = GT row
// (KG rows per lane), detections visited in the order list.  The reference's inner loop over the
// GTs sorted "ignored last" keeps the best IoU >= min(t, 1 - 1e-10) among the still unmatched
// rows of the detection's category, the LAST one on ties, and never looks at ignored rows once a
// counted row matched: that is "max over counted rows, else max over ignored rows, ties to the
// higher row", a wave reduction.
template <int KG>
__global__ __launch_bounds__(WAVE) void coco_match_kernel(
    const float4 *__restrict__ dt_boxes, const int32_t *__restrict__ dt_labels, const int32_t *__restrict__ order,
    const int32_t *__restrict__ count, const int32_t *__restrict__ rank, const float4 *__restrict__ gt_boxes,
    const int32_t *__restrict__ gt_labels, int R, int G, CocoParams P, int32_t *__restrict__ out_dt_match,
    uint8_t *__restrict__ out_dt_ignore, int32_t *__restrict__ out_gt_match, uint8_t *__restrict__ out_gt_ignore)
{
    const int lane = (int)threadIdx.x;
    const uint32_t at = (uint32_t)(P.A * P.T);
    const uint32_t b = blockIdx.x / at;
    const uint32_t rem = blockIdx.x - b * at;
    const uint32_t a = rem / (uint32_t)P.T, t = rem - a * (uint32_t)P.T;
    const double lo = P.lo[a], hi = P.hi[a];
    const double thr0 = fmin(P.thr[t], 1.0 - 1e-10);                      // cocoeval.py:280

    float gx[KG], gy[KG], gw[KG], gh[KG];
    int glab[KG], gig[KG], gm[KG];
#pragma unroll
    for (int c = 0; c < KG; ++c) {
        const int g = c * WAVE + lane;
        glab[c] = -1;
        gig[c] = 0;
        gm[c] = 0;
        gx[c] = gy[c] = gw[c] = gh[c] = 0.0f;
        if (g < G) {
            const int32_t l = gt_labels[(size_t)b * G + g];
            const float4 bx = gt_boxes[(size_t)b * G + g];
            gx[c] = bx.x;
            gy[c] = bx.y;
            gw[c] = bx.z - bx.x;                                          // evaluate.py:40
            gh[c] = bx.w - bx.y;                                          // :39
            const double area = (double)(gh[c] * gw[c]);                  // :41 fp32 product, widened
            if (l >= 0 && l < P.C) {
                glab[c] = l;
                gig[c] = (area < lo || area > hi) ? 1 : 0;                // cocoeval.py:255
            }
            if (t == 0 && out_gt_ignore)
                out_gt_ignore[((size_t)b * P.A + a) * G + g] = glab[c] < 0 ? (uint8_t)255 : (uint8_t)gig[c];
        }
    }

    const size_t dt_out = (((size_t)b * P.A + a) * P.T + t) * (size_t)R;
    const int n = count[b];
    for (int j0 = 0; j0 < n; j0 += WAVE) {
        const int my = j0 + lane;
        const int slot = my < n ? order[(size_t)b * R + my] : -1;
        float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
        int dlab = -2;
        if (slot >= 0) {
            db = dt_boxes[(size_t)b * R + slot];
            dlab = dt_labels[(size_t)b * R + slot];
        }
        const float dw = db.z - db.x, dh = db.w - db.y;                   // evaluate.py:19-20
        const double darea = (double)dw * (double)dh;                     // :177
        int res_m = 0;
        int res_ig = (darea < lo || darea > hi) ? 1 : 0;                  // cocoeval.py:302-303 (if unmatched)
        const int cnt = min(WAVE, n - j0);
        for (int j = 0; j < cnt; ++j) {
            const int lab = __builtin_amdgcn_readlane(dlab, j);
            bool any = false;
#pragma unroll
            for (int c = 0; c < KG; ++c) any = any || (glab[c] == lab && gm[c] == 0);
            if (__ballot(any) == 0ull) continue;
            const float x = lane_f(db.x, j), y = lane_f(db.y, j), w = lane_f(dw, j), h = lane_f(dh, j);
            double bi0 = -1.0, bi1 = -1.0;
            int bg0 = -1, bg1 = -1;
#pragma unroll
            for (int c = 0; c < KG; ++c) {
                if (glab[c] == lab && gm[c] == 0) {
                    const double iou = bb_iou(x, y, w, h, gx[c], gy[c], gw[c], gh[c]);
                    if (iou >= thr0) {
                        if (gig[c]) {
                            if (iou >= bi1) { bi1 = iou; bg1 = c * WAVE + lane; }
                        } else {
                            if (iou >= bi0) { bi0 = iou; bg0 = c * WAVE + lane; }
                        }
                    }
                }
            }
            const bool use1 = __ballot(bg0 >= 0) == 0ull;                 // no counted row matched: try ignored rows
            const double bi = use1 ? bi1 : bi0;
            const int bg = use1 ? bg1 : bg0;
            const unsigned long long mm = __ballot(bg >= 0);
            if (mm == 0ull) continue;
            int win;
            if (__popcll(mm) == 1) {
                win = __ffsll(mm) - 1;
            } else {
                double mx = bi;
#pragma unroll
                for (int off = 1; off < WAVE; off <<= 1) mx = fmax(mx, __shfl_xor(mx, off));
                int gs = (bg >= 0 && bi == mx) ? bg : -1;
#pragma unroll
                for (int off = 1; off < WAVE; off <<= 1) gs = max(gs, __shfl_xor(gs, off));
                win = __builtin_amdgcn_readfirstlane(gs) & (WAVE - 1);
            }
            const int wg = __builtin_amdgcn_readlane(bg, win);
            const int dslot = __builtin_amdgcn_readlane(slot, j);
#pragma unroll
            for (int c = 0; c < KG; ++c)
                if (lane == win && c == (wg >> 6)) gm[c] = dslot + 1;     // cocoeval.py:300
            if (lane == j) {
                res_m = wg + 1;                                           // :299
                res_ig = use1 ? 1 : 0;                                    // :298
            }
        }
        if (slot >= 0) {
            out_dt_match[dt_out + slot] = res_m;
            out_dt_ignore[dt_out + slot] = (uint8_t)res_ig;
        }
    }
    for (int s = lane; s < R; s += WAVE)
        if (rank[(size_t)b * R + s] < 0) {
            out_dt_match[dt_out + s] = 0;
            out_dt_ignore[dt_out + s] = 0;
        }
    if (out_gt_match) {
        const size_t gt_out = (((size_t)b * P.A + a) * P.T + t) * (size_t)G;
#pragma unroll
        for (int c = 0; c < KG; ++c) {
            const int g = c * WAVE + lane;
            if (g < G) out_gt_match[gt_out + g] = gm[c];
        }
    }
}

}  // namespace

extern "C" {

int odbox_coco_records(const float *boxes, size_t n, float *out_xywh, float *out_area_f32, double *out_area_f64,
                       void *stream)
{
    if (n == 0) return 0;
    if (!boxes) return fail(ODBOX_EINVAL, "odbox_coco_records: NULL boxes");
    if (!aligned16(boxes) || (out_xywh && !aligned16(out_xywh)))
        return fail(ODBOX_EINVAL, "odbox_coco_records: box tensors must be 16-byte aligned");
    const size_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_coco_records: too many boxes");
    hipLaunchKernelGGL(coco_records_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes, n, (float4 *)out_xywh, out_area_f32, out_area_f64);
    return launch_status("coco_records");
}

int odbox_coco_iou(const float *dt_boxes, const float *gt_boxes, int B, int R, int G, double *out_iou, void *stream)
{
    if (B < 0 || R < 0 || G < 0) return fail(ODBOX_EINVAL, "odbox_coco_iou: negative shape");
    if (B == 0 || R == 0 || G == 0) return 0;
    if (!dt_boxes || !gt_boxes || !out_iou) return fail(ODBOX_EINVAL, "odbox_coco_iou: NULL pointer");
    if (!aligned16(dt_boxes) || !aligned16(gt_boxes))
        return fail(ODBOX_EINVAL, "odbox_coco_iou: box tensors must be 16-byte aligned");
    const size_t total = (size_t)B * R * G;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_coco_iou: grid too large");
    hipLaunchKernelGGL(coco_iou_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)dt_boxes, (const float4 *)gt_boxes, R, G, out_iou, total);
    return launch_status("coco_iou");
}

size_t odbox_coco_match_workspace(int B, int R)
{
    if (B < 1 || R < 1) return 0;
    return (((size_t)B * R + (size_t)B) * sizeof(int32_t) + 255) / 256 * 256;
}

int odbox_coco_match(const float *dt_boxes, const float *dt_scores, const int32_t *dt_labels, const float *gt_boxes,
                     const int32_t *gt_labels, int B, int R, int G, int C, const double *iou_thrs_h, int T,
                     const double *area_rng_h, int A, int max_det, int32_t *out_dt_match, uint8_t *out_dt_ignore,
                     int32_t *out_gt_match, uint8_t *out_gt_ignore, int32_t *out_dt_rank, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    if (B < 0 || R < 0 || G < 0 || C < 1 || max_det < 1)
        return fail(ODBOX_EINVAL, "odbox_coco_match: bad shape B=%d R=%d G=%d C=%d max_det=%d", B, R, G, C, max_det);
    if (T < 1 || T > COCO_MAX_T || A < 1 || A > COCO_MAX_A || !iou_thrs_h || !area_rng_h)
        return fail(ODBOX_EINVAL, "odbox_coco_match: need 1..%d IoU thresholds and 1..%d area ranges (host arrays)",
                    COCO_MAX_T, COCO_MAX_A);
    if (R > ODBOX_MAX_RET) return fail(ODBOX_EINVAL, "odbox_coco_match: R=%d exceeds %d", R, ODBOX_MAX_RET);
    if (G > ODBOX_COCO_MAX_GT) return fail(ODBOX_EINVAL, "odbox_coco_match: G=%d exceeds %d", G, ODBOX_COCO_MAX_GT);
    if (B == 0 || R == 0) return 0;
    if (!dt_boxes || !dt_scores || !dt_labels || !out_dt_match || !out_dt_ignore || !out_dt_rank)
        return fail(ODBOX_EINVAL, "odbox_coco_match: NULL pointer");
    if (G > 0 && (!gt_boxes || !gt_labels)) return fail(ODBOX_EINVAL, "odbox_coco_match: NULL ground truth");
    if (!aligned16(dt_boxes) || (G > 0 && !aligned16(gt_boxes)))
        return fail(ODBOX_EINVAL, "odbox_coco_match: box tensors must be 16-byte aligned");
    const size_t need = odbox_coco_match_workspace(B, R);
    if (!workspace || workspace_bytes < need)
        return fail(ODBOX_ENOSPC, "odbox_coco_match: workspace %zu bytes, need %zu", workspace_bytes, need);
    const uint64_t waves = (uint64_t)B * (uint64_t)A * (uint64_t)T;
    if (waves > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_coco_match: grid too large");

    CocoParams P;
    P.T = T;
    P.A = A;
    P.C = C;
    for (int i = 0; i < COCO_MAX_T; ++i) P.thr[i] = i < T ? iou_thrs_h[i] : 2.0;
    for (int i = 0; i < COCO_MAX_A; ++i) {
        P.lo[i] = i < A ? area_rng_h[2 * i] : 0.0;
        P.hi[i] = i < A ? area_rng_h[2 * i + 1] : 0.0;
    }
    int32_t *order = (int32_t *)workspace;
    int32_t *count = order + (size_t)B * R;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(coco_order_kernel, dim3((uint32_t)B), dim3(ORDER_THREADS), (size_t)R * 9, st, dt_scores, dt_labels,
                       R, C, max_det, order, count, out_dt_rank);
    if (int rc = launch_status("coco_order")) return rc;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)waves), dim3(WAVE), 0, st, (const float4 *)dt_boxes, dt_labels,
                           (const int32_t *)order, (const int32_t *)count, (const int32_t *)out_dt_rank,
                           (const float4 *)gt_boxes, gt_labels, R, G, P, out_dt_match, out_dt_ignore, out_gt_match,
                           out_gt_ignore);
    };
    if (G <= 64) launch(coco_match_kernel<1>);
    else if (G <= 128) launch(coco_match_kernel<2>);
    else if (G <= 256) launch(coco_match_kernel<4>);
    else if (G <= 512) launch(coco_match_kernel<8>);
    else launch(coco_match_kernel<16>);
    return launch_status("coco_match");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// ground-truth preparation, the step that feeds the matcher.
// detection.py:97-141 (prepare_example, the box arithmetic of the evaluation path: square-pad
// offset, corner polygon, rescale to the network input size, cast, polygon2bbox :77-95, cut at
// max_items_in_image) and :257-262 (padded_batch: rows beyond an image's count are -1).
// Images arrive ragged (CSR row offsets), leave as the dense [B, max_items, 4] / [B, max_items]
// tensors the matcher reads.  One thread per output row.
// ------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float round_half(float v) { return __half2float(__float2half_rn(v)); }

// polygon2bbox over the 4 corners, in the tensor's dtype (fp16 values stay exactly representable
// through min/max; want_wh subtracts in that dtype too)
template <bool HALF>
__device__ __forceinline__ float4 poly_bbox(const float px[4], const float py[4], int mode)
{
    const float xmin = fminf(fminf(px[0], px[1]), fminf(px[2], px[3]));
    const float ymin = fminf(fminf(py[0], py[1]), fminf(py[2], py[3]));
    const float xmax = fmaxf(fmaxf(px[0], px[1]), fmaxf(px[2], px[3]));
    const float ymax = fmaxf(fmaxf(py[0], py[1]), fmaxf(py[2], py[3]));
    if (mode == 1) return make_float4(ymin, xmin, ymax, xmax);                          // want_yx
    if (mode == 2) {                                                                     // want_wh
        if (HALF)
            return make_float4(xmin, ymin, __half2float(__hsub(__float2half_rn(xmax), __float2half_rn(xmin))),
                               __half2float(__hsub(__float2half_rn(ymax), __float2half_rn(ymin))));
        return make_float4(xmin, ymin, xmax - xmin, ymax - ymin);
    }
    return make_float4(xmin, ymin, xmax, ymax);
}

// FROM_XYWH: boxes are the dataset's (x0, y0, w, h) in original pixels and sizes[b] = (height, width)
// of the original image (detection.py:98-123, :131 with the evaluation path's unchanged canvas).
// else: boxes are [rows, 4, 2] corner polygons on a canvas of side sizes[b] (the output of the
// training augmentation, detection.py:126,131).
template <bool FROM_XYWH, bool HALF>
__global__ __launch_bounds__(256) void gt_prepare_kernel(const float *__restrict__ src, const int32_t *__restrict__ labels,
                                                         const int32_t *__restrict__ row_offsets, int rows,
                                                         const float *__restrict__ sizes, int B, int max_items,
                                                         float image_size, int mode, float4 *__restrict__ out_boxes,
                                                         int32_t *__restrict__ out_labels)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)B * (uint32_t)max_items) return;
    const uint32_t b = i / (uint32_t)max_items, k = i - b * (uint32_t)max_items;
    // offsets come from device memory: clamped into [0, rows] so that a corrupted table gives wrong
    // rows, never an access outside the two ragged arrays
    const int32_t r0 = min(max(row_offsets[b], 0), rows), r1 = min(max(row_offsets[b + 1], r0), rows);
    float4 box = make_float4(-1.f, -1.f, -1.f, -1.f);                    // padding_values -1 (:262)
    int32_t lab = -1;
    if ((int32_t)k < r1 - r0) {                                           // rows past max_items are cut (:140-141)
        const size_t r = (size_t)r0 + k;
        float px[4], py[4], cur;
        if (FROM_XYWH) {
            const float4 s = reinterpret_cast<const float4 *>(src)[r];
            const float H = sizes[2 * b], W = sizes[2 * b + 1];
            const float mx = fmaxf(H, W);                                 // :103
            const float x0 = s.x + (mx - W) / 2.0f;                      // :110
            const float y0 = s.y + (mx - H) / 2.0f;                      // :111
            px[0] = x0;       py[0] = y0;                                 // :118-121
            px[1] = x0 + s.z; py[1] = y0;
            px[2] = x0 + s.z; py[2] = y0 + s.w;
            px[3] = x0;       py[3] = y0 + s.w;
            cur = (float)(int)mx;                                         // :104-109,131: side of the padded canvas
        } else {
            const float4 a = reinterpret_cast<const float4 *>(src)[2 * r], c = reinterpret_cast<const float4 *>(src)[2 * r + 1];
            px[0] = a.x; py[0] = a.y; px[1] = a.z; py[1] = a.w;
            px[2] = c.x; py[2] = c.y; px[3] = c.z; py[3] = c.w;
            cur = sizes[b];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            px[q] = (px[q] / cur) * image_size;                           // :135
            py[q] = (py[q] / cur) * image_size;
            if (HALF) {                                                   // :136 cast to the image dtype
                px[q] = round_half(px[q]);
                py[q] = round_half(py[q]);
            }
        }
        box = poly_bbox<HALF>(px, py, mode);                              // :138
        lab = labels[r];
    }
    out_boxes[i] = box;
    out_labels[i] = lab;
}

int gt_prepare(const char *name, bool from_xywh, const float *src, const int32_t *labels, const int32_t *row_offsets,
               int rows, const float *sizes, int B, int max_items, float image_size, int half_round, int mode, float *out_boxes,
               int32_t *out_labels, void *stream)
{
    if (B < 0 || max_items < 0 || rows < 0) return fail(ODBOX_EINVAL, "%s: negative shape", name);
    if (B == 0 || max_items == 0) return 0;
    if (!row_offsets || !sizes || !out_boxes || !out_labels) return fail(ODBOX_EINVAL, "%s: NULL pointer", name);
    if (rows > 0 && (!src || !labels)) return fail(ODBOX_EINVAL, "%s: NULL boxes / labels with rows = %d", name, rows);
    if (mode < 0 || mode > 2) return fail(ODBOX_EINVAL, "%s: mode must be 0 (xyxy), 1 (yxyx) or 2 (xywh)", name);
    if ((src && !aligned16(src)) || !aligned16(out_boxes))
        return fail(ODBOX_EINVAL, "%s: box tensors must be 16-byte aligned", name);
    if (!(image_size > 0.0f)) return fail(ODBOX_EINVAL, "%s: image_size must be positive", name);
    const uint64_t total = (uint64_t)B * (uint64_t)max_items;
    if (total > 0x7fffffffull) return fail(ODBOX_EINVAL, "%s: too many rows", name);
    const dim3 grid((uint32_t)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, block, 0, st, src, labels, row_offsets, rows, sizes, B, max_items, image_size, mode,
                           (float4 *)out_boxes, out_labels);
    };
    if (from_xywh) { if (half_round) launch(gt_prepare_kernel<true, true>); else launch(gt_prepare_kernel<true, false>); }
    else           { if (half_round) launch(gt_prepare_kernel<false, true>); else launch(gt_prepare_kernel<false, false>); }
    return launch_status(name);
}

// ------------------------------------------------------------------------------------------
// infdet.py:140-174  detections of the padded, resized square back to original image pixels:
//   max_side = max(imh, imw); pad = (max_side - im{h,w}) / 2; square_scale = max_side / image_size
//   bb.astype(float) * square_scale, then x -= pad_x, y -= pad_y          (all Python floats: fp64)
// and the per-image loop stops at the first score <= 0 (:161-162).  One block per image.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void restore_boxes_kernel(const float4 *__restrict__ boxes,
                                                            const float *__restrict__ scores,
                                                            const int32_t *__restrict__ orig_hw, int R, double image_size,
                                                            double *__restrict__ out, int32_t *__restrict__ out_count)
{
    __shared__ int s_first;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_first = R;
    __syncthreads();
    int first = R;
    for (int r = threadIdx.x; r < R; r += 256)
        if (scores[(size_t)b * R + r] <= 0.0f) { first = r; break; }        // `if score <= 0: break`
    if (first < R) atomicMin(&s_first, first);
    __syncthreads();
    const int count = s_first;
    const double imh = (double)orig_hw[2 * b], imw = (double)orig_hw[2 * b + 1];
    const double max_side = imh > imw ? imh : imw;
    const double pad_y = (max_side - imh) / 2.0, pad_x = (max_side - imw) / 2.0;
    const double square_scale = max_side / image_size;
    for (int r = threadIdx.x; r < R; r += 256) {
        double4 o = make_double4(0.0, 0.0, 0.0, 0.0);
        if (r < count) {
            const float4 v = boxes[(size_t)b * R + r];
            o = make_double4((double)v.x * square_scale - pad_x, (double)v.y * square_scale - pad_y,
                             (double)v.z * square_scale - pad_x, (double)v.w * square_scale - pad_y);
        }
        double *dst = out + ((size_t)b * R + r) * 4;
        dst[0] = o.x; dst[1] = o.y; dst[2] = o.z; dst[3] = o.w;
    }
    if (threadIdx.x == 0) out_count[b] = count;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU plumbing (no counterpart in the reference): the fixed-size records one
// all-gather moves.  [n,4] boxes + [n] scores + [n] int32 labels <-> [n,6] words, label bit-cast.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_records_kernel(const float4 *__restrict__ boxes,
                                                           const float *__restrict__ scores,
                                                           const int32_t *__restrict__ labels, uint32_t n,
                                                           float *__restrict__ rec)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 b = boxes[i];
    float *r = rec + (size_t)i * 6;
    r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
    r[4] = scores[i];
    r[5] = __int_as_float(labels[i]);
}

__global__ __launch_bounds__(256) void unpack_records_kernel(const float *__restrict__ rec, uint32_t n,
                                                             float4 *__restrict__ boxes, float *__restrict__ scores,
                                                             int32_t *__restrict__ labels)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float *r = rec + (size_t)i * 6;
    boxes[i] = make_float4(r[0], r[1], r[2], r[3]);
    scores[i] = r[4];
    labels[i] = __float_as_int(r[5]);
}

// ------------------------------------------------------------------------------------------
// the reference's legacy / offline IoU conventions, NumPy float64 arithmetic.
//   mode 0  coco.py:169-191 calc_iou: (cx, cy, h, w) boxes, +1 pixel on the intersection's sides only,
//           the second operand's area handed in (np_anchor_areas);
//   mode 1  calculate_metrics.py:22-45 calc_ious: xyxy, the same +1, areas from the corners.
// Same association as the reference: w = (xx1 - xx0) + 1, union = (area_a + area_b) - inter.
// np.maximum(0, x) is fmax(0, x) for the finite values of the contract.
// ------------------------------------------------------------------------------------------
struct LegacyEdges {
    double x0, x1, y0, y1, area;
};

template <int MODE>
__device__ __forceinline__ LegacyEdges legacy_edges(const double *__restrict__ b, const double *__restrict__ area_in)
{
    LegacyEdges e;
    if (MODE == 0) {
        const double hw = b[3] / 2, hh = b[2] / 2;
        e.x0 = b[0] - hw; e.x1 = b[0] + hw;
        e.y0 = b[1] - hh; e.y1 = b[1] + hh;
        e.area = area_in ? *area_in : b[2] * b[3];          // box[2] * box[3] for the first operand
    } else {
        e.x0 = b[0]; e.y0 = b[1]; e.x1 = b[2]; e.y1 = b[3];
        e.area = (b[2] - b[0]) * (b[3] - b[1]);
    }
    return e;
}

__device__ __forceinline__ double legacy_ovr(const LegacyEdges &a, const LegacyEdges &b)
{
    const double w = fmax(0.0, (fmin(a.x1, b.x1) - fmax(a.x0, b.x0)) + 1.0);
    const double h = fmax(0.0, (fmin(a.y1, b.y1) - fmax(a.y0, b.y0)) + 1.0);
    const double inter = w * h;
    return inter / ((a.area + b.area) - inter);
}

template <int MODE>
__global__ __launch_bounds__(256) void legacy_iou_kernel(const double *__restrict__ boxes_a, const double *__restrict__ boxes_b,
                                                         const double *__restrict__ area_b, uint32_t K, uint32_t N,
                                                         double *__restrict__ out)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const LegacyEdges b = legacy_edges<MODE>(boxes_b + (size_t)n * 4, MODE == 0 ? area_b + n : nullptr);
    for (uint32_t k = blockIdx.y; k < K; k += gridDim.y) {
        const LegacyEdges a = legacy_edges<MODE>(boxes_a + (size_t)k * 4, nullptr);     // wave-uniform: scalar loads
        out[(size_t)k * N + n] = legacy_ovr(a, b);
    }
}

// coco.py:379-441, the sequential per-GT matcher, one thread per (image, anchor): the GT rows are
// visited in order; an anchor is (re)assigned when iou > thr and iou > its running maximum, which the
// reference keeps in a float32 array (:379) -- so the maximum is stored ROUNDED and widened again for
// the next comparison, and that is reproduced here.  Rows with w <= 3 or h <= 3 are skipped (:418);
// the library's -1 padding rows fall under that rule.
constexpr int LEGACY_GT_CHUNK = 256;

__global__ __launch_bounds__(256) void legacy_match_kernel(const double *__restrict__ anchors, const double *__restrict__ areas,
                                                           const double *__restrict__ gt_xywh, const int32_t *__restrict__ gt_labels,
                                                           uint32_t N, uint32_t G, double thr, double *__restrict__ out_boxes,
                                                           double *__restrict__ out_labels, float *__restrict__ out_max_iou,
                                                           double *__restrict__ out_label_sum)
{
    __shared__ double s_gt[LEGACY_GT_CHUNK][4];      // (cx, cy, h, w) of the rows that pass the size filter
    __shared__ double s_lab[LEGACY_GT_CHUNK];
    __shared__ int s_n;
    __shared__ double s_sum[4];
    const uint32_t b = blockIdx.y, n = blockIdx.x * 256u + threadIdx.x;
    const bool live = n < N;
    double box[4] = {0.0, 0.0, 0.0, 0.0}, label = 0.0;
    float best = 0.0f;
    LegacyEdges an{};
    if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) box[q] = anchors[(size_t)n * 4 + q];                // true_bboxes = anchors.copy()
        an = legacy_edges<0>(box, areas + n);
    }
    for (uint32_t g0 = 0; g0 < G; g0 += LEGACY_GT_CHUNK) {
        __syncthreads();
        if (threadIdx.x == 0) {                      // order-preserving compaction of the chunk's usable rows
            int cnt = 0;
            const uint32_t g1 = min(G, g0 + (uint32_t)LEGACY_GT_CHUNK);
            for (uint32_t g = g0; g < g1; ++g) {
                const double *r = gt_xywh + ((size_t)b * G + g) * 4;
                const double x0 = r[0], y0 = r[1], w = r[2], h = r[3];
                if (w <= 3.0 || h <= 3.0) continue;
                s_gt[cnt][0] = x0 + w / 2;           // :424-427
                s_gt[cnt][1] = y0 + h / 2;
                s_gt[cnt][2] = h;
                s_gt[cnt][3] = w;
                s_lab[cnt] = (double)gt_labels[(size_t)b * G + g];
                ++cnt;
            }
            s_n = cnt;
        }
        __syncthreads();
        const int cnt = s_n;
        if (live)
            for (int g = 0; g < cnt; ++g) {
                const LegacyEdges e = legacy_edges<0>(s_gt[g], nullptr);
                const double iou = legacy_ovr(e, an);
                if (iou > thr && iou > (double)best) {                                   // :390-392
#pragma unroll
                    for (int q = 0; q < 4; ++q) box[q] = s_gt[g][q];
                    label = s_lab[g];
                    best = (float)iou;                                                   // :413 into the float32 array
                }
            }
    }
    if (live) {
        const size_t o = (size_t)b * N + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) out_boxes[o * 4 + q] = box[q];
        out_labels[o] = label;
        if (out_max_iou) out_max_iou[o] = best;
    }
    if (out_label_sum) {                             // true_labels.sum() of :439 (labels are integers: exact in any order)
        double s = live ? label : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
            if (t != 0.0) atomicAdd(out_label_sum + b, t);
        }
    }
}

__global__ __launch_bounds__(256) void zero_f64_kernel(double *__restrict__ p, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace

extern "C" {

int odbox_legacy_iou(const double *boxes_a, const double *boxes_b, const double *area_b, int K, int N, int mode,
                     double *out_iou, void *stream)
{
    if (K < 0 || N < 0) return fail(ODBOX_EINVAL, "odbox_legacy_iou: negative shape K=%d N=%d", K, N);
    if (mode != 0 && mode != 1) return fail(ODBOX_EINVAL, "odbox_legacy_iou: mode must be 0 (cx,cy,h,w) or 1 (xyxy)");
    if (K == 0 || N == 0) return 0;
    if (!boxes_a || !boxes_b || !out_iou || (mode == 0 && !area_b)) return fail(ODBOX_EINVAL, "odbox_legacy_iou: NULL pointer");
    if (((uintptr_t)boxes_a | (uintptr_t)boxes_b | (uintptr_t)area_b | (uintptr_t)out_iou) & 7u)
        return fail(ODBOX_EINVAL, "odbox_legacy_iou: tensors must be 8-byte aligned");
    const dim3 grid(((uint32_t)N + 255u) / 256u, (uint32_t)(K < 1024 ? K : 1024)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0)
        hipLaunchKernelGGL(legacy_iou_kernel<0>, grid, block, 0, st, boxes_a, boxes_b, area_b, (uint32_t)K, (uint32_t)N, out_iou);
    else
        hipLaunchKernelGGL(legacy_iou_kernel<1>, grid, block, 0, st, boxes_a, boxes_b, area_b, (uint32_t)K, (uint32_t)N, out_iou);
    return launch_status("odbox_legacy_iou");
}

int odbox_legacy_match(const double *anchors_cxcyhw, const double *anchor_areas, const double *gt_xywh,
                       const int32_t *gt_labels, int B, int N, int G, double iou_thr, double *out_boxes,
                       double *out_labels, float *out_max_iou, double *out_label_sum, void *stream)
{
    if (B < 0 || N < 0 || G < 0) return fail(ODBOX_EINVAL, "odbox_legacy_match: negative shape B=%d N=%d G=%d", B, N, G);
    if (B == 0 || N == 0) return 0;
    if (B > 65535) return fail(ODBOX_EINVAL, "odbox_legacy_match: B=%d over the limit of 65535", B);
    if (!anchors_cxcyhw || !anchor_areas || !out_boxes || !out_labels || (G > 0 && (!gt_xywh || !gt_labels)))
        return fail(ODBOX_EINVAL, "odbox_legacy_match: NULL pointer");
    if (((uintptr_t)anchors_cxcyhw | (uintptr_t)anchor_areas | (uintptr_t)gt_xywh | (uintptr_t)out_boxes |
         (uintptr_t)out_labels | (uintptr_t)out_label_sum) & 7u)
        return fail(ODBOX_EINVAL, "odbox_legacy_match: fp64 tensors must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (out_label_sum) {
        hipLaunchKernelGGL(zero_f64_kernel, dim3(((uint32_t)B + 255u) / 256u), dim3(256), 0, st, out_label_sum, (uint32_t)B);
        if (int rc = launch_status("odbox_legacy_match (zero)")) return rc;
    }
    hipLaunchKernelGGL(legacy_match_kernel, dim3(((uint32_t)N + 255u) / 256u, (uint32_t)B), dim3(256), 0, st, anchors_cxcyhw,
                       anchor_areas, gt_xywh, gt_labels, (uint32_t)N, (uint32_t)G, iou_thr, out_boxes, out_labels, out_max_iou,
                       out_label_sum);
    return launch_status("odbox_legacy_match");
}

int odbox_gt_prepare_xywh(const float *boxes_xywh, const int32_t *labels, const int32_t *row_offsets, int rows,
                          const float *orig_hw, int B, int max_items, float image_size, int half_round, int mode,
                          float *out_boxes, int32_t *out_labels, void *stream)
{
    return gt_prepare("odbox_gt_prepare_xywh", true, boxes_xywh, labels, row_offsets, rows, orig_hw, B, max_items, image_size,
                      half_round, mode, out_boxes, out_labels, stream);
}

int odbox_gt_prepare_poly(const float *polys, const int32_t *labels, const int32_t *row_offsets, int rows,
                          const float *canvas_size, int B, int max_items, float image_size, int half_round, int mode, float *out_boxes,
                          int32_t *out_labels, void *stream)
{
    return gt_prepare("odbox_gt_prepare_poly", false, polys, labels, row_offsets, rows, canvas_size, B, max_items, image_size,
                      half_round, mode, out_boxes, out_labels, stream);
}

int odbox_restore_boxes(const float *boxes, const float *scores, const int32_t *orig_hw, int B, int R, int image_size,
                        double *out_boxes, int32_t *out_count, void *stream)
{
    if (B < 0 || R < 0 || image_size < 1)
        return fail(ODBOX_EINVAL, "odbox_restore_boxes: bad shape B=%d R=%d image_size=%d", B, R, image_size);
    if (B == 0) return 0;
    if (!orig_hw || !out_count || (R > 0 && (!boxes || !scores || !out_boxes)))
        return fail(ODBOX_EINVAL, "odbox_restore_boxes: NULL pointer");
    if (((uintptr_t)boxes & 15u) || ((uintptr_t)out_boxes & 7u))
        return fail(ODBOX_EINVAL, "odbox_restore_boxes: boxes must be 16-byte aligned, out_boxes 8-byte");
    hipLaunchKernelGGL(restore_boxes_kernel, dim3((uint32_t)B), dim3(256), 0, (hipStream_t)stream, (const float4 *)boxes,
                       scores, orig_hw, R, (double)image_size, out_boxes, out_count);
    return launch_status("restore_boxes");
}

int odbox_pack_records(const float *boxes, const float *scores, const int32_t *labels, size_t n, float *records,
                       void *stream)
{
    if (n == 0) return 0;
    if (!boxes || !scores || !labels || !records) return fail(ODBOX_EINVAL, "odbox_pack_records: NULL pointer");
    if (!aligned16(boxes)) return fail(ODBOX_EINVAL, "odbox_pack_records: boxes must be 16-byte aligned");
    if (n > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_pack_records: too many records");
    hipLaunchKernelGGL(pack_records_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes, scores, labels, (uint32_t)n, records);
    return launch_status("pack_records");
}

int odbox_unpack_records(const float *records, size_t n, float *boxes, float *scores, int32_t *labels, void *stream)
{
    if (n == 0) return 0;
    if (!boxes || !scores || !labels || !records) return fail(ODBOX_EINVAL, "odbox_unpack_records: NULL pointer");
    if (!aligned16(boxes)) return fail(ODBOX_EINVAL, "odbox_unpack_records: boxes must be 16-byte aligned");
    if (n > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_unpack_records: too many records");
    hipLaunchKernelGGL(unpack_records_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, records,
                       (uint32_t)n, (float4 *)boxes, scores, labels);
    return launch_status("unpack_records");
}

}  // extern "C"
