// odbox_train.hip -- training-side kernels (anchor tiling, fused IoU matcher + target encoder,
// dense IoU, elementwise box ops) + their C-ABI (include/odbox.h), gfx950 / wave64.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).  -ffp-contract=off is
// part of the contract: the reference evaluates every TensorFlow op as a separately rounded
// fp32 op, so no multiply-add here may be fused into an FMA; divisions are IEEE (hipcc's
// default correctly-rounded fp32 divide).  No MFMA: the path is elementwise and reduction work,
// HBM-bound except the dense-crowd matcher (VALU-bound).
//
// Reference citations are file:line under the reference tree (bioothod/object_detection).
#include "odbox_common.h"
#include <atomic>
#include <mutex>
#include <type_traits>

namespace odbox {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const char *last_error() { return g_err; }

}  // namespace odbox

using namespace odbox;

static std::atomic<int> g_output_policy{ODBOX_POLICY_AUTO};     // odbox_set_output_policy

namespace {

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
    float ix0 = vmax(a.x, g.x);
    float ix1 = vmin(a.z, g.z);
    float iy0 = vmax(a.y, g.y);
    float iy1 = vmin(a.w, g.w);
    float i_area = vmax(ix1 - ix0, 0.0f) * vmax(iy1 - iy0, 0.0f);
    float u_area = ((a_area + g_area) - i_area) + 1e-8f;
    return i_area / u_area;
}

// ------------------------------------------------------------------------------------------
// Fused matcher + encoder: anchors.py:116-213 (+ :215-274, :277-299, bndbox.py:254-299).
//
// One block = one tile of ME_TILE anchors of one image.
//   phase 1  thread = anchor: GT boxes of the image staged in LDS (32-byte records: box, area, kind of row)
//            in chunks of ME_GCHUNK; per-wave culling of the rows that cannot overlap into a work list of
//            record offsets (branch-free, one LDS round trip per 64 rows); then the division-free loop over
//            the list, four rows per step: approximate quotients, the two largest kept, the exact IEEE
//            divisions only for the winner and for a chunk whose runner-up is inside the winner's error band
//            (first max wins, bit for bit the reference's argmax and IoU).
//   phase 2  thresholds, out-of-image test, label, offsets; compact outputs stored directly; the
//            5 regression floats staged in LDS.
//   phase 3  the two dense outputs of the tile are contiguous in memory ([tile,5] and
//            [tile,C+1] floats) and written as 16-byte stores over the FLAT element range, aligned
//            to cache lines (head/tail scalar).  The [tile,C+1] block is almost all zeros, and
//            zeros do not depend on the match: they are issued from INSIDE phase 1, right after its
//            global loads, and drain while the VALUs match; the non-zero elements (state, the
//            one 1.0) are patched per anchor afterwards.  This stream is the HBM-bound part:
//            4(C+1)+20 of the 4(C+1)+36 bytes per anchor.
// The #defines below are tuning knobs for A/B builds (ODBOX_EXTRA_HIPCC_FLAGS, tools/ab_variants.sh);
// the defaults are the measured winners.
// ------------------------------------------------------------------------------------------
#ifndef ODBOX_ME_TILE
#define ODBOX_ME_TILE 256                   // anchors per block in the loss kernels (A/B: 128 and 512 are slower)
#endif
constexpr int ME_TILE = ODBOX_ME_TILE;
#ifndef ODBOX_ME_TILE_ENC
#define ODBOX_ME_TILE_ENC 256               // match_encode's tile when the dense output is small (tuning: -D)
#endif
constexpr int ME_TILE_ENC = ODBOX_ME_TILE_ENC;
#ifndef ODBOX_ZERO_ALIGN
#define ODBOX_ZERO_ALIGN 128               // bytes the zero stream's vector stores are aligned to (A/B: 16)
#endif
constexpr uint32_t ZERO_ALIGN = ODBOX_ZERO_ALIGN;
#ifndef ODBOX_LOSS_ALIGN
#define ODBOX_LOSS_ALIGN 128               // bytes the loss kernels' score / gradient vectors are aligned to (A/B: 16)
#endif
constexpr uint32_t LOSS_ALIGN = ODBOX_LOSS_ALIGN;
#ifndef ODBOX_ZERO_LEAD
#define ODBOX_ZERO_LEAD 2                  // zero-stream vectors per thread issued before the matcher's loads (same-box A/B on config 1 / 2: 0 -> 22.4 / 34.5 us, 2 -> 21.7 / 34.0, 4 -> 22.1 / 34.1, 8 -> 22.4 / 34.0 and config 3 +3 %)
#endif
constexpr int ZERO_LEAD = ODBOX_ZERO_LEAD;
#ifndef ODBOX_STREAM_MB
#define ODBOX_STREAM_MB 512                // dense label bytes beyond which match_encode bypasses the Infinity Cache
#endif
#ifndef ODBOX_ME_TILE_STREAM
#define ODBOX_ME_TILE_STREAM 64                // same-box A/B, 128 -> 64: configs[2] share 358 -> 347 us, configs[1] on cold buffers 45.3 -> 43.6 us
#endif
constexpr int ME_TILE_STREAM = ODBOX_ME_TILE_STREAM;      // match_encode with a dense output far beyond the Infinity Cache
#ifndef ODBOX_ME_GCHUNK
#define ODBOX_ME_GCHUNK 512
#endif
constexpr int ME_GCHUNK = ODBOX_ME_GCHUNK;      // GT rows staged in LDS at a time
#ifndef ODBOX_FAST_ROWS
#define ODBOX_FAST_ROWS 4                  // GT rows per step of the division-free matcher loop
#endif
constexpr int FAST_ROWS = ODBOX_FAST_ROWS;
// work-list entries of match_tile: 16 bits hold every offset and cost one extraction per row; 32-bit entries cost none but
// 4 KB more LDS per 256-anchor block (same-box A/B, configs[3]: 288-290 us against 300-303 us; the fused loss forward
// 41.2 against 43.0 us)
#ifdef ODBOX_LIST32
typedef uint32_t ListEntry;
#else
typedef uint16_t ListEntry;
#endif

// Phases 1 and 2 of the matcher for the calling thread's anchor n0 + threadIdx.x of image b:
// anchors.py:215-274 (argmax IoU, thresholds), :178-184 (out-of-image), :187-194 (label).
// Shared by match_encode_kernel and match_loss_kernel; must be called by all ME_TILE threads.
struct TileMatch {
    float4 a;          // the anchor
    float best_iou;    // max_g IoU (tf.reduce_max)
    int best;          // argmax_g IoU (tf.argmax: first max)
    int state;         // -1 ignore / 0 background / 1 foreground
    int label;         // class of the matched GT if foreground and in [0, C), else -1
    bool live;         // n < N
};

// `early` is called once by every thread after the block's first global loads (anchor, first GT
// chunk) have been consumed and before any of the matching arithmetic: work issued there (the
// dense zero stream of match_encode_kernel) drains while the VALUs do the match.
// REDUCE_LATE: the wave's bounding-box reduction runs after `early` (so it, too, overlaps what was
// issued there: 34.5 vs 34.9 us for match+encode on config 2) instead of before the GT staging
// (better for the loss kernel, whose prefetched score load is in flight meanwhile: 45.0 vs 45.9 us).
//
// GTBEST (match_encode_kernel only): the per-GT best anchor, an EXTRA output that changes nothing of
// the above (north_star's "per-GT best-anchor" reduction; the reference only has the diagnostic of
// coco.py:399-402 and never forces a label from it).  For GT row g of image b: the anchor with the
// largest IoU over all N anchors, ties to the LOWEST anchor (numpy / tf argmax over the anchor axis).
// The block keeps one 64-bit key (ordered IoU bits, ~anchor) per staged GT row in LDS, seeded with the
// row's key in gbest_keys[b, g] as some earlier block left it (a device-scope load; stale = low, never
// wrong).  In the division-free loop a row costs one broadcast LDS read and one compare per lane:
// "can my approximate quotient, at the top of its error band, reach the row's best so far".  Only a
// wave with such a lane goes on: DPP / readlane reduction of the quotient over the wavefront, the
// lanes within the band of the wave's maximum (one, unless anchors tie) take the exact IEEE division
// and fold their key into the row's LDS maximum (kept in the staged row's record).  Per chunk each row some lane posted for
// goes to gbest_keys[b, g] with one global atomicMax.  Pairs with IoU exactly +0 (culled rows
// included) post nothing: a row nobody posted for has IoU +0 everywhere (or -1: padding) and its
// argmax is anchor 0 -- filled in by gt_best_finalize_kernel, which also recomputes the winner's
// IoU, so the value carries the exact bits of IoU(anchor, g), sign of zero included.  Rows that
// cannot be culled (negative-area GT rows or anchors: values below zero are possible) take the exact
// loop and post every pair.
__device__ __forceinline__ unsigned long long gbest_key(float v, uint32_t n)
{
    return ((unsigned long long)f32_ordered(v) << 32) | (0xffffffffu - n);
}

// Diagnostic builds only (-DODBOX_COUNT_PAIRS, tools/count_pairs.py): how many (anchor, GT row) pairs the matcher
// really evaluates -- [0] pairs of live lanes in the work lists, [1] 64-row culling steps, [2] waves -- which is what
// the VALU roofline of the dense-crowd configuration is priced on (bench.py, match_encode_cfg3).
#ifdef ODBOX_COUNT_PAIRS
__device__ unsigned long long g_pair_count[4];
#endif

template <int TILE, bool REDUCE_LATE, bool GTBEST = false, typename EarlyWork>
__device__ __forceinline__ TileMatch match_tile(const float4 *__restrict__ anchors,
                                                const float4 *__restrict__ gt_boxes,
                                                const int32_t *__restrict__ gt_labels, uint32_t b, uint32_t n0, int N,
                                                int G, int C, float img_h, float img_w, float neg_thr, float pos_thr,
                                                EarlyWork early, unsigned long long *gbest_keys = nullptr)
{
    __shared__ uint32_t s_dirty[GTBEST ? ME_GCHUNK / 32 : 1];      // GTBEST: rows some lane of this block posted a key for
    // one staged GT row: box, area, kind (0 = a box with a finite area >= 0, 1 = padded row, 2 = any other area: negative,
    // infinite, NaN) and, for GTBEST, the row's best key so far -- 32 bytes, so that one LDS address serves the box
    // (ds_read_b128), the area (ds_read_b32 at +16) and the key's upper word (ds_read_b32 at +28)
    struct __align__(16) GtRow { float4 box; float area; float kind; unsigned long long key; };
    __shared__ GtRow s_row[ME_GCHUNK];
    // per wave: the GT rows it must evaluate, as BYTE OFFSETS into s_row (row * 32: the matcher loop spends no VALU
    // instruction on addresses, and the best row is carried as its offset), + slack for the read-ahead
    __shared__ __align__(16) ListEntry s_list[TILE / WAVE][ME_GCHUNK + FAST_ROWS];
    static_assert(sizeof(GtRow) == 32 && (size_t)ME_GCHUNK * sizeof(GtRow) <= (size_t)(ListEntry)~(ListEntry)0 + 1, "list entries are byte offsets of 32-byte rows");
    auto row_at = [&](uint32_t off) -> const GtRow & { return *reinterpret_cast<const GtRow *>(reinterpret_cast<const char *>(s_row) + off); };
    constexpr int GCHUNK_WORDS = ME_GCHUNK / 32;
    // GTBEST: fold a key into the staged row's maximum and mark the row for the chunk's global atomicMax
    auto post_key = [&](int g, unsigned long long key) {
        atomicMax(&s_row[g].key, key);
        atomicOr(&s_dirty[g >> 5], 1u << (g & 31));
    };
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = n0 + tid;
    const bool live = n < (uint32_t)N;
    const float4 a = live ? anchors[n] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float a_area = (a.w - a.y) * (a.z - a.x);                       // bndbox.py:285

    // Bounding box of this wave's 64 anchors (and their smallest area).  A GT row that does not
    // reach into it has intersection 0 with every lane; with non-negative areas its IoU is then
    // exactly +0 (union >= 1e-8), so the row can be left out of the loop and accounted for at
    // the end as "first skipped zero row".  Padded rows (IoU -1 for every anchor) are skipped
    // the same way.  Dead lanes (n >= N) do not widen the box.
    float wx0 = live ? a.x : INFINITY, wy0 = live ? a.y : INFINITY;
    float wx1 = live ? a.z : -INFINITY, wy1 = live ? a.w : -INFINITY;
    // every live anchor of the wave has a finite area >= 0 (false for a negative, infinite or NaN area: the
    // comparisons are ordered ones and the unit is built with NaNs honoured)
    const bool can_skip_zero = __ballot(live && !(a_area >= 0.0f && a_area < INFINITY)) == 0ull;
    // over the wavefront by DPP -- quads, half rows, rows, then row 0 into 1 and 2 into 3, then 1 into 2 and 3: lane 63 holds
    // the four results -- 24 VALU instructions and no LDS round trip (the butterfly of __shfl_xor was 24 ds_bpermute in six
    // dependent round trips at the head of every block's critical path).  One statement, the four chains interleaved:
    // a DPP read of a VGPR needs two wait states after its write, and every chain's next step comes four instructions later.
    auto reduce_box = [&] {
#define ODBOX_BOX_STEP(ctrl)                                        \
        "v_min_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n"          \
        "v_min_f32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n"          \
        "v_max_f32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n"          \
        "v_max_f32_dpp %3, %3, %3 " ctrl " bank_mask:0xf\n"
        asm volatile("s_nop 1\n"
                     ODBOX_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf")
                     ODBOX_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf")
                     ODBOX_BOX_STEP("row_half_mirror row_mask:0xf")
                     ODBOX_BOX_STEP("row_mirror row_mask:0xf")
                     ODBOX_BOX_STEP("row_bcast:15 row_mask:0xa")
                     ODBOX_BOX_STEP("row_bcast:31 row_mask:0xc")
                     "s_nop 1"
                     : "+v"(wx0), "+v"(wy0), "+v"(wx1), "+v"(wy1));
#undef ODBOX_BOX_STEP
        wx0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx0), 63));
        wy0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy0), 63));
        wx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx1), 63));
        wy1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy1), 63));
    };
    if (!REDUCE_LATE) reduce_box();

    // ---- phase 1: argmax_g IoU ------------------------------------------------------------
    int best = 0;
    float best_iou = -INFINITY;
    int first_zero = -1, first_pad = -1;                // wave-uniform: first skipped rows
    const float4 *gtb = gt_boxes + (size_t)b * G;
    for (int g0 = 0; g0 < G; g0 += ME_GCHUNK) {
        const int cnt = min(ME_GCHUNK, G - g0);
        if (g0) __syncthreads();
        for (int g = tid; g < cnt; g += TILE) {
            float4 gb = gtb[g0 + g];
            float area = (gb.w - gb.y) * (gb.z - gb.x);                  // bndbox.py:286
            float gmax = vmax(vmax(gb.x, gb.y), vmax(gb.z, gb.w));    // bndbox.py:295
            s_row[g].box = gb;
            s_row[g].area = area;
            s_row[g].kind = gmax < 0.0f ? 1.0f : (area >= 0.0f && area < INFINITY) ? 0.0f : 2.0f;
            if (GTBEST) {
                // device scope: a plain load would be served by this XCD's L2, which never sees the other XCDs' atomics
                const unsigned long long seed = __hip_atomic_load(&gbest_keys[(size_t)b * G + g0 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_row[g].key = seed;
                if (g < GCHUNK_WORDS) s_dirty[g] = 0u;
            }
        }
        // every global load issued so far has been consumed: from here to the label / box gathers at
        // the end the match is VALU and LDS work only, so memory work issued now overlaps it
        if (g0 == 0) {
            early();
            if (REDUCE_LATE) reduce_box();
        }
        __syncthreads();
        // this wave's work list: 64 GT rows per step, one per lane, order preserved
        uint32_t nlist = 0;
        uint64_t odd_rows = 0ull;                          // listed rows whose area is negative, infinite or NaN
        // (branch-free on purpose: written with an `if (g < cnt)` around the reads, hipcc made three dependent LDS round
        // trips and four exec-mask branches of every step; the clamped row of a lane beyond the chunk is read and ignored)
        for (int gs = 0; gs < cnt; gs += WAVE) {
            const int g = gs + (int)lane;
            const GtRow &row = s_row[min(g, cnt - 1)];
            const float4 gb = row.box;
            const float kind = row.kind;
            // the wave's masks as 64-bit integers built from the elementary compares (a ballot of a combined predicate
            // makes hipcc materialise it in a VGPR and compare it again); `keep` per lane for the store below
            const bool in = g < cnt, k0 = kind == 0.0f, k1 = kind == 1.0f;
            const bool outside = (gb.x >= wx1) | (gb.z <= wx0) | (gb.y >= wy1) | (gb.w <= wy0);
            const uint64_t m_in = __ballot(in);
            const uint64_t m_out = __ballot(gb.x >= wx1) | __ballot(gb.z <= wx0) | __ballot(gb.y >= wy1) | __ballot(gb.w <= wy0);
            const uint64_t mp = m_in & __ballot(k1);
            const uint64_t mz = can_skip_zero ? m_in & __ballot(k0) & m_out : 0ull;
            const uint64_t mk = m_in & ~mp & ~mz;
            odd_rows |= mk & __ballot(kind == 2.0f);
            const bool keep = in & !k1 & !(k0 & can_skip_zero & outside);
            if (keep) s_list[wave][nlist + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = (ListEntry)(g * (int)sizeof(GtRow));
            nlist += (uint32_t)__popcll(mk);
            if (first_zero < 0 && mz != 0ull) first_zero = g0 + gs + __ffsll((unsigned long long)mz) - 1;
            if (first_pad < 0 && mp != 0ull) first_pad = g0 + gs + __ffsll((unsigned long long)mp) - 1;
        }
#ifdef ODBOX_COUNT_PAIRS
        const unsigned long long live_lanes = (unsigned long long)__popcll(__ballot(live));
        if (lane == 0) {
            atomicAdd(&g_pair_count[0], (unsigned long long)nlist * live_lanes);
            atomicAdd(&g_pair_count[1], (unsigned long long)((cnt + WAVE - 1) / WAVE));
            if (g0 == 0) atomicAdd(&g_pair_count[2], 1ull);
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- the usual case, division-free: every area in play is finite and >= 0, so every IoU is a number >= +0
        // (an intersection is at most the smaller area, hence finite; a sum of areas that overflows gives 0) --------
        // The correctly rounded division is 13 of the ~29 issue cycles of a pair, and all the argmax needs
        // from it is "which row has the greatest quotient".  q = i * rcp(u'), u' = ((a + 1e-8) + g) - i: the union
        // summed in another order than bndbox.py:289 (one addition less per row: a + 1e-8 is the lane's constant).
        // With u = 2^-24: i is the same number in both; no sum here cancels (a + g - i >= (a + g) / 2), so u' is within
        // 5u of the real union and the reference's rounded union within 4u; v_rcp_f32 is 1 ulp (2u), the product u:
        // q is within 13u of the reference's rounded quotient.  Two rows whose q differ by more than 2^-18 = 64u
        // relative therefore compare the same way as their exact quotients.  So the loop keeps the TWO largest q of
        // the chunk (and the exact best of the chunks before it) -- a compare, a select, one v_med3 and one v_max per
        // row, no band arithmetic -- and only when the runner-up ends inside that band of the winner (a tie, or next to
        // one: practically never) is the chunk's list walked again with the exact IEEE divisions.  i == 0 is the exact
        // quotient +0 and 0 is never inside the band of 0 (strict >, first maximum wins).  The best row's exact IoU is
        // computed once, after the loop.  Same argmax and the same value as the exact loop below
        // (tests/test_gpu_match.py::test_ties_and_near_ties_in_the_division_free_matcher).
        auto post_chunk = [&] {
            if (!GTBEST) return;
            __syncthreads();                                 // every wave's maxima of this chunk are in
            unsigned long long *dst = gbest_keys + (size_t)b * G + g0;
            for (int g = tid; g < cnt; g += TILE) {
                if ((s_dirty[g >> 5] >> (g & 31)) & 1u) atomicMax(&dst[g], s_row[g].key);
            }
        };
        if (can_skip_zero && odd_rows == 0ull && nlist != 0u) {
            constexpr float BAND_HI = 1.0f + 0x1p-18f;
            const float a_eps = a_area + 1e-8f;
            auto exact_iou = [&](int row) {
                const float4 bb = row >= g0 ? s_row[row - g0].box : gtb[row];
                return pair_iou(a, a_area, bb, (bb.w - bb.y) * (bb.z - bb.x));
            };
            // best so far as an approximate quotient (-inf = none yet, 0 = an exact zero) and the runner-up
            float qb = best_iou, q2 = -INFINITY;
            const int best_in = best;
            int best_off = -1;                                 // the chunk's best row as its offset in s_row, -1 = none beats `best`
            const float best_iou_in = best_iou;
            // R rows at a time.  The wave runs beside five or six others and its own chain of LDS round trips (list entry
            // -> box -> arithmetic, row after row as hipcc schedules it when left alone) is what bounds the loop, not the
            // VALU count: so the R boxes and areas are read back to back behind a scheduling barrier -- one round trip
            // per step -- and the step's list entries arrive with the step before (`g`, read ahead by the caller).
            auto rows = [&](const int *g, auto rc) {
                constexpr int R = decltype(rc)::value;
                float ia[R], ua[R], q[R];
                float4 gb[R];
                float g_area[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    gb[r] = row_at(g[r]).box;
                    g_area[r] = row_at(g[r]).area;
                }
                __builtin_amdgcn_sched_barrier(0);
                // (two rows per v_pk_add_f32 / v_pk_mul_f32 were tried for the subtractions, products and sums: same-box
                // A/B on configs[3] 352-358 us packed against 340-346 us plain -- the register pairing costs more moves
                // than the packing saves)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float ix0 = vmax(a.x, gb[r].x), ix1 = vmin(a.z, gb[r].z);
                    const float iy0 = vmax(a.y, gb[r].y), iy1 = vmin(a.w, gb[r].w);
                    ia[r] = vmax(ix1 - ix0, 0.0f) * vmax(iy1 - iy0, 0.0f);
                    ua[r] = (a_eps + g_area[r]) - ia[r];            // bndbox.py:289 with the 1e-8 added first (see above), > 0 here
                    q[r] = ia[r] * __builtin_amdgcn_rcpf(ua[r]);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    best_off = q[r] > qb ? g[r] : best_off;                // strict: the first of equal quotients stays
                    q2 = __builtin_amdgcn_fmed3f(q[r], qb, q2);            // q2 <= qb: the median is the new runner-up
                    qb = vmax(qb, q[r]);
                }
                if (GTBEST) {
                    // per-GT best anchor: can any lane reach the row's best so far?  The key's upper word is the
                    // ordered IoU of a value > 0 (sign bit set on top of the float's bits), or 0 = nothing yet
                    bool reach[R];
                    bool any_reach = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        // (|x|: the sign bit is the ordered form's marker of a value >= 0; as a source modifier it costs nothing)
                        const float cur = fabsf(__uint_as_float(reinterpret_cast<const uint32_t *>(&row_at(g[r]).key)[1]));
                        reach[r] = ia[r] > 0.0f && q[r] * BAND_HI >= cur;
                        any_reach = any_reach || reach[r];
                    }
                    if (__ballot(any_reach && live) != 0ull) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (__ballot(reach[r] && live) == 0ull) continue;
                            // the wave's largest quotient (all >= 0: their bits order like integers)
                            float m = reach[r] && live ? q[r] : 0.0f;
                            m = vmax(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, true)));    // quad_perm [1,0,3,2]
                            m = vmax(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, true)));    // quad_perm [2,3,0,1]
                            m = vmax(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x141, 0xf, 0xf, true)));   // row_half_mirror
                            m = vmax(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x140, 0xf, 0xf, true)));   // row_mirror
                            const uint32_t mb = max(max((uint32_t)__builtin_amdgcn_readlane(__float_as_int(m), 0),
                                                        (uint32_t)__builtin_amdgcn_readlane(__float_as_int(m), 16)),
                                                    max((uint32_t)__builtin_amdgcn_readlane(__float_as_int(m), 32),
                                                        (uint32_t)__builtin_amdgcn_readlane(__float_as_int(m), 48)));
                            const float qmax = __uint_as_float(mb);
                            // a lane below the band of the wave's maximum has a strictly smaller exact IoU than the lane
                            // that holds it; the others (one lane, unless anchors tie) divide exactly and post
                            if (reach[r] && live && q[r] * BAND_HI >= qmax) {
                                const float v = ia[r] / (((a_area + g_area[r]) - ia[r]) + 1e-8f);
                                if (v > 0.0f) post_key(g[r] / (int)sizeof(GtRow), gbest_key(v, n));
                            }
                        }
                    }
                }
            };
            uint32_t k = 0;
#ifdef ODBOX_EXP_ROWS_SHIFT
            nlist >>= ODBOX_EXP_ROWS_SHIFT;
#endif
            if (nlist >= (uint32_t)FAST_ROWS) {
                int gn[FAST_ROWS];
#pragma unroll
                for (int r = 0; r < FAST_ROWS; ++r) gn[r] = s_list[wave][r];
                for (; k + FAST_ROWS <= nlist; k += FAST_ROWS) {
                    int g[FAST_ROWS];
#pragma unroll
                    for (int r = 0; r < FAST_ROWS; ++r) g[r] = gn[r];
                    // the next step's entries (the list has FAST_ROWS slots of slack behind its last row: read, never used)
#pragma unroll
                    for (int r = 0; r < FAST_ROWS; ++r) gn[r] = s_list[wave][k + FAST_ROWS + r];
                    rows(g, std::integral_constant<int, FAST_ROWS>{});
                }
            }
            for (; k < nlist; ++k) {
                const int g1 = s_list[wave][k];
                rows(&g1, std::integral_constant<int, 1>{});
            }
            if (__ballot(q2 * BAND_HI > qb) != 0ull) {         // wave-uniform, practically never: this chunk again, exactly
                best = best_in;
                best_iou = best_iou_in;
                for (k = 0; k < nlist; ++k) {
                    const int g = (int)(s_list[wave][k] / sizeof(GtRow));
                    const float v = pair_iou(a, a_area, s_row[g].box, s_row[g].area);
                    if (v > best_iou) { best_iou = v; best = g0 + g; }
                }
            } else {
                if (best_off >= 0) best = g0 + best_off / (int)sizeof(GtRow);
                best_iou = exact_iou(best);
            }
            post_chunk();
            continue;
        }
#pragma unroll 2
        for (uint32_t k = 0; k < nlist; ++k) {
            const int g = (int)(s_list[wave][k] / sizeof(GtRow));
            const float4 gb = s_row[g].box;
            const float2 ga = make_float2(s_row[g].area, s_row[g].kind);
            const float v = pair_iou(a, a_area, gb, ga.x);               // never a padded row here
            if (v > best_iou) { best_iou = v; best = g0 + g; }           // tf.argmax: first max
            if (GTBEST) {
                const bool forced = !can_skip_zero || !(ga.x >= 0.0f);
                if (live && (v > 0.0f || forced)) post_key(g, gbest_key(v, n));
            }
        }
        post_chunk();
    }
    // fold the skipped rows back in: value 0 at first_zero, value -1 at first_pad; ties go to
    // the lower row like tf.argmax
    if (first_zero >= 0 && (0.0f > best_iou || (0.0f == best_iou && first_zero < best))) {
        best_iou = 0.0f;
        best = first_zero;
    }
    if (first_pad >= 0 && (-1.0f > best_iou || (-1.0f == best_iou && first_pad < best))) {
        best_iou = -1.0f;
        best = first_pad;
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
    TileMatch r;
    r.a = a;
    r.best_iou = best_iou;
    r.best = best;
    r.state = state;
    r.label = label;
    r.live = live;
    return r;
}

// STREAM: the dense labels leave through cache-bypassing (nontemporal) stores.  Measured: outputs
// that fit the 256 MB Infinity Cache (cfg 2: 234 MB) are 1.5x FASTER with plain stores, outputs far
// beyond it (cfg 3: 2.2 GB per GPU) 6 % faster bypassing it.  The streaming variant also runs
// 128-anchor tiles (cfg 3), the small launches prefer 256 (cfg 2: 128 and 512 are both slower).
template <bool STREAM, bool GTBEST>
__global__ __launch_bounds__(STREAM ? ME_TILE_STREAM : ME_TILE_ENC) void match_encode_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ gt_boxes,
    const int32_t *__restrict__ gt_labels, int N, int G, int C, float img_h, float img_w,
    float neg_thr, float pos_thr, float *__restrict__ out_reg, float *__restrict__ out_lab,
    int32_t *__restrict__ out_idx, int8_t *__restrict__ out_state, float *__restrict__ out_iou,
    unsigned long long *gbest_keys, uint32_t tiles_per_image, uint32_t remap)
{
    constexpr int TILE = STREAM ? ME_TILE_STREAM : ME_TILE_ENC;
    __shared__ float s_reg[TILE * 5];

    const uint32_t tid = threadIdx.x;
    // neighbouring tiles on one XCD (shared L2) pays when the output streams to HBM (config 3: 358 vs
    // 388 us); while it fits the Infinity Cache the plain order is 1-3 % faster (config 2: 33.3 vs 33.8 us)
    // ... and both orders run back to front: the coarse pyramid levels at the end of the anchor array
    // have the longest work lists (every GT row reaches into their tiles), so they go first and the
    // launch ends on cheap tiles (config 4: 413 -> 400 us, configs 1 / 3: -1.5 % / -1 %)
    // remap == 2 (many GT rows, small output): tile-major like GTBEST below -- the coarse tiles of ALL images first.  Their
    // waves walk every GT row (ten times the average list) and live ten times as long; image-major order starts the
    // last image's coarse tiles at the very end of the launch and the chip then waits for a few waves that run alone
    // (configs[3]: the launch took the same 340 us with 11 % fewer VALU instructions and with a third of the LDS
    // round trips -- only the average occupancy fell; tools/exp_cfg4_per_launch.py, profiles/README.md)
    const bool tile_major = GTBEST || remap == 2u;
    const uint32_t blk = gridDim.x - 1u - (remap == 1u && !GTBEST ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x);
    // GTBEST: tile-major order (all images' last tile, then all images' tile before it, ...), so that the tiles of one
    // image run one after the other rather than all at once: a tile then finds the per-GT keys its predecessors posted
    // and most of its (wave, row) pairs fail the "can I reach the best so far" test instead of reducing over the
    // wavefront (dense crowd with the output on: 700 us with all tiles of an image in flight together).  XCD = block
    // mod 8 = image mod 8: balanced, and an image's GT rows and keys stay in one L2.
    const uint32_t n_img = gridDim.x / tiles_per_image;
    const uint32_t b = tile_major ? blk % n_img : blk / tiles_per_image;
    // (GTBEST: `remap` carries a multiplier coprime to tiles_per_image -- the tiles of an image in a scattered order: a
    // sweep through the anchor grid approaches every GT box step by step, and every step raises its best)
    // (the scatter is worth a third of the launch and its details a tenth: configs[3] with the per-GT output 382 us; the
    // same multiplier walked from the other end 421 us; the coarse tiles first -- they walk every GT row -- and the rest
    // scattered 455 us; tiles in order 540 us)
    const uint32_t n0 = (GTBEST ? (uint32_t)(((uint64_t)(blk / n_img) * remap) % tiles_per_image)
                                : tile_major ? blk / n_img : blk - b * tiles_per_image) * TILE;
    const uint32_t n = n0 + tid;
    const bool live = n < (uint32_t)N;
    const uint32_t tile_n = min((uint32_t)TILE, (uint32_t)N - n0);

    // ---- phase 3b first: the [tile,C+1] one-hot + state rows are zero except for the state of
    // not-background anchors and the one 1.0 of foreground anchors.  The zeros do not depend on the
    // match, so they are streamed (flat 16-byte stores) from inside match_tile, right after its
    // global loads, and drain while the VALUs do the matching; the few non-zero elements are
    // patched by their anchor's thread afterwards, behind an explicit s_waitcnt vmcnt(0) and the block
    // barrier: the patch overwrites bytes of this stream.
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const uint32_t d = (uint32_t)C + 1u;
    float *lab_p = out_lab ? out_lab + ((size_t)b * N + n0) * d : nullptr;
    // `first` = the opening slice, issued before the matcher's own loads (they queue behind it, so it
    // is kept short: ZERO_LEAD vectors per thread); the rest goes out from inside match_tile.
    auto zero_stream = [&](bool first) {
        if (!out_lab) return;
        // vectors start on a 128-byte line (up to 31 scalar head elements), so every wave-wide store
        // covers 8 whole cache lines: with tiles that begin anywhere (N not a multiple of 32) the
        // 16-byte-aligned version wrote two partial lines per instruction
        const uint32_t count = tile_n * d;
        const uint32_t head = min(count, (uint32_t)((ZERO_ALIGN - ((uintptr_t)lab_p & (ZERO_ALIGN - 1u))) & (ZERO_ALIGN - 1u)) >> 2);
        const uint32_t nvec = (count - head) >> 2;
        const v4f_ z = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t lead = min(nvec, (uint32_t)ZERO_LEAD * TILE);
        if (!first && tid < head) lab_p[tid] = 0.0f;
        for (uint32_t v = (first ? 0u : lead) + tid; v < (first ? lead : nvec); v += TILE) {
            v4f_ *dst = reinterpret_cast<v4f_ *>(lab_p + head + 4 * v);
            // compiler-visible stores only: the patch below overwrites bytes of this stream.  (Device-scope
            // sc1 stores from inline asm win a bare fill -- 25.9 vs 27.9 us for config 2's labels,
            // tools/exp/fill_shapes.hip -- but lose inside this kernel on the config-2 grid, B 64: 37.2 vs
            // 34.4 us; `sc1 nt` gains 1 % on config 3.  Neither is worth a store the compiler cannot see.)
            if (STREAM) __builtin_nontemporal_store(z, dst);
            else *dst = z;
        }
        const uint32_t done = head + 4 * nvec;
        if (!first && tid < count - done) lab_p[done + tid] = 0.0f;
    };
    if (ZERO_LEAD > 0) zero_stream(true);

    const TileMatch tm = match_tile<TILE, true, GTBEST>(anchors, gt_boxes, gt_labels, b, n0, N, G, C, img_h, img_w, neg_thr,
                                                        pos_thr, [&] { zero_stream(false); }, gbest_keys);
    const float4 a = tm.a;
    const float best_iou = tm.best_iou;
    const int best = tm.best, state = tm.state, label = tm.label;
    const float4 *gtb = gt_boxes + (size_t)b * G;

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
    // Every zero store of this thread has reached L2 before any thread of the block patches (the
    // box gather above already implies it when out_reg is set: vmcnt counts in order).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // s_reg complete; the zero stream of the whole tile is in L2

    // ---- phase 3b, second half: the non-zero elements of this thread's own row ----------------
    if (out_lab && live) {
        float *rowp = lab_p + (size_t)tid * d;
        if (state != 0) rowp[d - 1] = (float)state;
        if (label >= 0) rowp[label] = 1.0f;
    }

    // ---- phase 3a: [tile,5] regression rows, flat 16-byte stores ---------------------------
    if (out_reg) {
        float *p = out_reg + ((size_t)b * N + n0) * 5;
        const uint32_t count = tile_n * 5;
        const uint32_t head = min(count, (uint32_t)((ZERO_ALIGN - ((uintptr_t)p & (ZERO_ALIGN - 1u))) & (ZERO_ALIGN - 1u)) >> 2);
        const uint32_t nvec = (count - head) >> 2;
        if (tid < head) p[tid] = s_reg[tid];
        for (uint32_t v = tid; v < nvec; v += TILE) {
            const uint32_t e = head + 4 * v;
            *reinterpret_cast<float4 *>(p + e) = make_float4(s_reg[e], s_reg[e + 1], s_reg[e + 2], s_reg[e + 3]);
        }
        const uint32_t done = head + 4 * nvec;
        if (tid < count - done) p[done + tid] = s_reg[done + tid];
    }
}

// ------------------------------------------------------------------------------------------
// match_encode_sweep_kernel: the same outputs as match_encode_kernel<true, false>, written in the order
// cold HBM likes best (anchors.py:116-213; same expressions, same bits).
//
// A store stream into COLD memory is fastest as a chip-wide lockstep sweep: every block writes one 4 KB
// piece per step and the 512 pieces of a step are contiguous (a 2 MB window that moves through the
// tensor): 35 us for configs[1]'s 221 MB of labels against 40-41 us for any tile-shaped pattern
// (tools/exp/fill_shapes.hip, profiles/README.md).  The sweep keeps that rate only while its waves do
// nothing else, so the block is specialised (tools/exp/sweep_spec.hip):
//   SWEEP_SW store waves   piece after piece of zeros, nothing else; each publishes in LDS how many of its
//                          pieces have reached L2 (stores complete in order: s_waitcnt vmcnt(K) leaves K
//                          in flight, no drain).
//   SWEEP_MW matcher waves lane = (one of four pieces, one of the <= 16 anchor rows that touch it): exact
//                          argmax over the image's G <= 64 GT rows, states, label, offsets; compact outputs
//                          and the regression row by the lane of the piece the row STARTS in; the non-zero
//                          label elements (state, the one 1.0) patched by the lane of the piece each
//                          element lies IN, once the store waves say that piece's zeros are out -- a row
//                          that straddles two pieces is matched twice, by the two blocks that own them.
// Taken by odbox_match_encode for streamed outputs (ODBOX_POLICY_STREAM; AUTO beyond ODBOX_STREAM_MB or for a buffer
// that may not be in the Infinity Cache any more, see ODBOX_POLICY_AUTO)
// with wide label rows (C + 1 >= 69: at most 16 rows per piece), G <= 64, 16-byte aligned labels of at most
// ODBOX_SWEEP_MAX_MB and no per-GT output; everything else stays with match_encode_kernel.
// ------------------------------------------------------------------------------------------
#ifndef ODBOX_SWEEP_MW
#define ODBOX_SWEEP_MW 12
#endif
#ifndef ODBOX_SWEEP_VMK
#define ODBOX_SWEEP_VMK 12
#endif
constexpr int SWEEP_PIECE = 1024;               // floats per piece
#ifndef ODBOX_SWEEP_SW
#define ODBOX_SWEEP_SW 4
#endif
constexpr int SWEEP_SW = ODBOX_SWEEP_SW;        // store waves: 1 KB each per piece
constexpr int SWEEP_MW = ODBOX_SWEEP_MW;        // matcher waves
constexpr int SWEEP_RPP = 16;                   // anchor rows that can touch one piece
constexpr int SWEEP_VMK = ODBOX_SWEEP_VMK;      // stores a store wave keeps in flight behind its progress counter
constexpr int SWEEP_MAX_G = 64;
constexpr int STREAM_AUTO_MAX_G = 80;           // GT rows up to which ODBOX_POLICY_AUTO streams large label tensors
constexpr int STREAM_MIN_ROW = 64;              // label floats per anchor from which odbox_match_encode streams at all (see there)
#ifndef ODBOX_SWEEP_MAX_MB
// label bytes up to which the sweep is taken.  Same-box A/B against the 64-anchor tiles on cold buffers (tools/exp_sweep.py):
// 135 MB (configs[0]) 27.2 -> 23.4 us, 234 MB (configs[1]) 43.2 -> 39.1 us, 270 MB 49.5 -> 43.5 us, 540 MB 90-93 -> 87-88 us,
// 1.08 GB 167-172 -> 179 us, 2.16 GB (configs[2] share) 327-332 -> 372-376 us: long launches amortise the tiles' ramp and
// reach 6.6-6.9 TB/s, which the sweep's narrow window does not.
#define ODBOX_SWEEP_MAX_MB 768
#endif

constexpr int SWEEP_LDS_ROWS = 3072;            // GT rows a block keeps in LDS (60 KB, two blocks per CU): the first SWEEP_LDS_ROWS / B of every image
constexpr int SWEEP_MAX_B = 8192;               // one 16-bit row bound per image in LDS

// The matcher waves first look at the whole batch's GT rows once (one row per lane, G <= 64), while the store waves
// already sweep: per image the BOUND = index behind its last valid row -- everything from there on is padding, which
// the argmax loop never visits (the reference pads every image to 64 rows, detection.py:246-266, of which a handful
// are boxes) and accounts for as one candidate of IoU -1 at index `bound`, exactly what visiting them in order gives
// under strict > -- and the first gs = min(G, SWEEP_LDS_ROWS / B) rows of every image go to LDS with their labels.
// Under the store stream a global round trip costs microseconds: the loop over LDS rows pays none (configs[1] into
// cold buffers 39.1 us; with every row from global memory 39.9-40.7 us; 64-row padded GT before the bound: 55 us).
__global__ __launch_bounds__(WAVE *(SWEEP_SW + SWEEP_MW)) void match_encode_sweep_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ gt_boxes, const int32_t *__restrict__ gt_labels, int N,
    int G, int C, float img_h, float img_w, float neg_thr, float pos_thr, float *__restrict__ out_reg, float *out_lab,
    int32_t *__restrict__ out_idx, int8_t *__restrict__ out_state, float *__restrict__ out_iou, uint32_t rows,
    uint32_t total, uint32_t pieces, uint32_t B, uint32_t gs, uint32_t xcd_runs)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    extern __shared__ float4 s_gt[];                    // [B * gs] boxes, [B * gs] labels, [B] bounds
    __shared__ volatile uint32_t s_done[SWEEP_SW];      // store wave w: the zeros of this block's pieces < s_done[w] are in L2
    __shared__ uint32_t s_staged;                       // matcher waves whose share of the GT rows is in LDS
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t d = (uint32_t)C + 1u;
    if (tid < SWEEP_SW) s_done[tid] = 0u;
    if (tid == SWEEP_SW) s_staged = 0u;
    __syncthreads();
    // xcd_runs: XCD x (blocks x, x + 8, ...) owns a contiguous eighth of every step's window, so neighbouring pieces
    // share an L2 -- the regression rows of a piece end inside cache lines whose other half the next piece's owner
    // writes, and two L2s each wrote their half back.  Same-box A/B: configs[0] 23.5 -> 22.8 us, configs[1] 39.55 ->
    // 38.7 us, 64-row padded GT 40.2 -> 39.4 us; 517 MB of labels 90.1 -> 91.2 us (hence the host's size rule)
    const uint32_t slot = xcd_runs ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (slot >= pieces) return;
    const uint32_t mine = (pieces - slot + gridDim.x - 1u) / gridDim.x;      // pieces of this block: slot + i * gridDim.x
    if (wave < SWEEP_SW) {
        __builtin_amdgcn_s_setprio(3);
        const v4f_ z = {0.0f, 0.0f, 0.0f, 0.0f};
        constexpr uint32_t PER = SWEEP_PIECE / SWEEP_SW / 256u;       // 16-byte stores per lane and piece
        // pieces that SWEEP_VMK stores in flight can belong to, plus one: a wave whose share of the tensor's last
        // piece lies beyond its end issues nothing for that piece
        constexpr uint32_t LAG = ((uint32_t)SWEEP_VMK + PER - 1u) / PER + 1u;
        static_assert(PER >= 1u && PER * 256u * SWEEP_SW == SWEEP_PIECE, "store waves must tile a piece with whole wave-wide stores");
        const uint32_t off = wave * (SWEEP_PIECE / SWEEP_SW) + lane * 4u;
        for (uint32_t i = 0; i < mine; ++i) {
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t f = (slot + i * gridDim.x) * SWEEP_PIECE + off + k * 256u;
                if (f + 3u < total) {
                    __builtin_nontemporal_store(z, reinterpret_cast<v4f_ *>(out_lab + f));
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < 3u; ++e)
                        if (f + e < total) out_lab[f + e] = 0.0f;
                }
            }
            // stores complete in order: all but the last SWEEP_VMK of this wave have reached L2
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SWEEP_VMK) : "memory");
            if (lane == 0u && i >= LAG) s_done[wave] = i + 1u - LAG;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0u) s_done[wave] = mine;
        return;
    }
    int32_t *s_lab = reinterpret_cast<int32_t *>(s_gt + (size_t)B * gs);
    uint16_t *s_bound = reinterpret_cast<uint16_t *>(s_lab + (size_t)B * gs);
    const uint32_t mw = wave - SWEEP_SW;
    {
        // K images per wave and trip, dealt round-robin over the matcher waves; a trip's loads go out together, lane = GT
        // row.  Same-box A/B at configs[1], K = 1 / 2 / 3 / 8: 39.4 / 39.4 / 39.7 / 39.85 us -- the trips are not what
        // costs (B = 64 needs six of them at K = 1), the registers the wider forms hold are
#ifndef ODBOX_SWEEP_STAGE_K
#define ODBOX_SWEEP_STAGE_K 2
#endif
        constexpr uint32_t K = ODBOX_SWEEP_STAGE_K;
        for (uint32_t b0 = mw; b0 < B; b0 += SWEEP_MW * K) {
            float4 row[K];
            int32_t lab[K];
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t bk = b0 + k * SWEEP_MW;
                const bool have = bk < B && lane < (uint32_t)G;
                const size_t at = (size_t)bk * G + lane;
                row[k] = have ? gt_boxes[at] : make_float4(-1.f, -1.f, -1.f, -1.f);
                lab[k] = have && gt_labels ? gt_labels[at] : -1;
            }
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t bk = b0 + k * SWEEP_MW;
                if (bk >= B) break;
                const bool pad = vmax(vmax(row[k].x, row[k].y), vmax(row[k].z, row[k].w)) < 0.0f;      // bndbox.py:295
                const uint64_t valid = __ballot(!pad && lane < (uint32_t)G);
                const uint32_t bound = valid ? 64u - (uint32_t)__clzll((long long)valid) : 0u;
                if (lane < min(bound, gs)) {
                    s_gt[bk * gs + lane] = row[k];
                    s_lab[bk * gs + lane] = lab[k];
                }
                if (lane == 0u) s_bound[bk] = (uint16_t)bound;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0u) atomicAdd(&s_staged, 1u);
        while (__hip_atomic_load(&s_staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)SWEEP_MW) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const uint32_t j = lane >> 4, rr = lane & 15u;
    // lane = (piece j of the batch, row rr of the piece): the row, or rows = none
    auto locate = [&](uint32_t t, uint32_t &i, uint32_t &f0, uint32_t &f1) -> uint32_t {
        i = t * 4u + j;
        if (i >= mine) return rows;
        f0 = (slot + i * gridDim.x) * SWEEP_PIECE;
        f1 = min(f0 + (uint32_t)SWEEP_PIECE, total);
        const uint32_t R = f0 / d + rr;                           // the piece's first row begins at or before f0
        return (R < rows && R * d < f1) ? R : rows;
    };
    uint32_t i_n, f0_n, f1_n;
    uint32_t R_n = locate(mw, i_n, f0_n, f1_n);
    float4 a_n = anchors[R_n < rows ? R_n % (uint32_t)N : 0u];
    for (uint32_t t = mw; t * 4u < mine; t += SWEEP_MW) {
        const uint32_t i = i_n, f0 = f0_n, f1 = f1_n, R = R_n;
        const float4 a = a_n;
        R_n = locate(t + SWEEP_MW, i_n, f0_n, f1_n);             // the next batch's anchors travel under this batch's match
        a_n = anchors[R_n < rows ? R_n % (uint32_t)N : 0u];
        if (R >= rows) continue;
        const uint32_t b = R / (uint32_t)N;
        const float a_area = (a.w - a.y) * (a.z - a.x);                       // bndbox.py:285
        const uint32_t gbase = b * (uint32_t)G, sbase = b * gs;
        const int bound = (int)s_bound[b];
        // ---- argmax_g IoU (anchors.py:253-256): padded rows count as -1, strict >, first maximum wins ----
        float best_iou = -INFINITY;
        int best = 0;
        auto four_rows = [&](int g0, int lim, auto load) {
            float4 gb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gb[k] = load(min(g0 + k, lim - 1));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float area = (gb[k].w - gb[k].y) * (gb[k].z - gb[k].x);                     // bndbox.py:286
                const bool pad = vmax(vmax(gb[k].x, gb[k].y), vmax(gb[k].z, gb[k].w)) < 0.0f;  // bndbox.py:295
                const float v = pad ? -1.0f : pair_iou(a, a_area, gb[k], area);
                if (g0 + k < lim && v > best_iou) { best_iou = v; best = g0 + k; }
            }
        };
        const int in_lds = min(bound, (int)gs) & ~3;             // whole groups of four from LDS, the rest from global memory
        for (int g0 = 0; g0 < in_lds; g0 += 4) four_rows(g0, in_lds, [&](int g) { return s_gt[sbase + (uint32_t)g]; });
        if (bound <= (int)gs) {
            if (in_lds < bound) four_rows(in_lds, bound, [&](int g) { return s_gt[sbase + (uint32_t)g]; });
        } else {
            for (int g0 = in_lds; g0 < bound; g0 += 4) four_rows(g0, bound, [&](int g) { return gt_boxes[gbase + (uint32_t)g]; });
        }
        // rows bound .. G-1 are padding: in order they are one candidate of -1 at `bound` (later equals never win)
        if (bound < G && -1.0f > best_iou) {
            best_iou = -1.0f;
            best = bound;
        }
        // ---- states, label (the second phase of match_tile, expression for expression) ----
        const bool positive = best_iou >= pos_thr;                            // anchors.py:267
        bool ignore = ((best_iou > neg_thr) && !positive) || (best_iou < 0.0f);   // :270-272
        const float xc = (a.x + a.z) / 2.0f;                                  // :178-184
        const float yc = (a.y + a.w) / 2.0f;
        ignore = ignore || (xc >= img_w) || (yc >= img_h);
        const int state = ignore ? -1 : (positive ? 1 : 0);
        int label = -1;
        if (positive && !ignore && gt_labels) label = best < (int)gs ? s_lab[sbase + (uint32_t)best] : gt_labels[gbase + (uint32_t)best];
        if (label < 0 || label >= C) label = -1;                              // tf.one_hot: zero row
        const uint32_t e0 = R * d;
        if (e0 >= f0) {                      // the row begins in this piece: its compact outputs and regression row are this lane's
            if (out_idx) out_idx[R] = best;
            if (out_state) out_state[R] = (int8_t)state;
            if (out_iou) out_iou[R] = best_iou;
            if (out_reg) {                                                    // anchors.py:286-297
                const float4 g = best < (int)gs && best < bound ? s_gt[sbase + (uint32_t)best] : gt_boxes[gbase + (uint32_t)best];
                const float aw = a.z - a.x, ah = a.w - a.y;
                // one 20-byte row per lane, 4-byte aligned: two stores (dwordx3 + dwordx2) instead of five
                struct __attribute__((packed, aligned(4))) Row5 { float v[5]; } r5;
                r5.v[0] = (((g.x - a.x) / aw) - 0.0f) / 0.2f;
                r5.v[1] = (((g.y - a.y) / ah) - 0.0f) / 0.2f;
                r5.v[2] = (((g.z - a.z) / aw) - 0.0f) / 0.2f;
                r5.v[3] = (((g.w - a.w) / ah) - 0.0f) / 0.2f;
                r5.v[4] = (float)state;
                *reinterpret_cast<Row5 *>(out_reg + (size_t)R * 5) = r5;
            }
        }
        // ---- the non-zero label elements of this row that lie in this piece, behind the piece's zeros ----
        const uint32_t ps = e0 + d - 1u;
        const uint32_t pl = e0 + (uint32_t)max(label, 0);
        const bool want_s = state != 0 && ps >= f0 && ps < f1;
        const bool want_l = label >= 0 && pl >= f0 && pl < f1;
        if (want_s || want_l) {
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int w = 0; w < SWEEP_SW; ++w) ok = ok && s_done[w] > i;
                if (!ok) __builtin_amdgcn_s_sleep(2);
            } while (!ok);
            if (want_s) out_lab[ps] = (float)state;
            if (want_l) out_lab[pl] = 1.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused matcher + loss: the consumer of the targets, metric.py:97-123
// (ModelMetric.__call__) with focal_loss (metric.py:9-38) and tf.keras.losses.Huber, evaluated
// straight from (match, state, offsets) -- the one-hot [B,N,C+1] tensor (90 % of the bytes of
// match+encode) and the [B,N,5] regression tensor are never written or read back.
//
//   class_sum = sum over not-ignored anchors and all C classes of
//                 -alpha_t * pow(1 - pt, gamma) * log(pt),  p = clip(pred, 1e-6, 1 - 1e-6),
//                 pt = y ? p : 1 - p,  alpha_t = y ? alpha : 1 - alpha,  y = (class == matched label)
//   dist_sum  = sum over foreground anchors of mean_4(huber_delta(pred_box - target_offsets))
//   npos      = number of foreground anchors (the reference's normaliser, metric.py:108-109)
//
// One block = the same 256-anchor tile as match_encode_kernel; phase 3 streams the tile's
// [tile, C] prediction rows as one flat run of 16-byte loads.  Per-block partial sums go to a
// small buffer and are added up in double by loss_reduce_kernel in a fixed order, so the result
// does not depend on scheduling.  Optional gradients of the UNNORMALISED sums w.r.t. the
// predictions (d class_sum / d pred_scores, d dist_sum / d pred_boxes); the caller divides by
// npos like the reference does (metric.py:121-123).
// ------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte score vectors in flight per thread in the loss kernels, same-box A/B on config 2
// (tools/ab_variants.sh): fp32 forward 44.9 / 45.0 / 44.0 / 48.0 us for 1 / 2 / 4 / 8 (backward 98.8 /
// 74.1 / 72.5 / 72.6); fp16 forward 38.9 / 38.9 / 41.0 / 45.5 (backward 52.3 / 51.9 / 55.1 / 61.0).
#ifdef ODBOX_LOSS_PREFETCH
constexpr int LOSS_PREFETCH_F32 = ODBOX_LOSS_PREFETCH, LOSS_PREFETCH_F16 = ODBOX_LOSS_PREFETCH;
#else
constexpr int LOSS_PREFETCH_F32 = 4, LOSS_PREFETCH_F16 = 2;
#endif

struct LossParams {
    float gamma, alpha, delta;
};

// One focal-loss element, branch-free.  This is floating-point work compared with a tolerance
// (1e-5 relative on the sums), so the transcendental parts use the hardware rate instructions:
// log2 through v_log_f32 (pt >= 1e-6, never denormal), pow(x, gamma) as x * sqrt(x) for the
// reference's gamma = 1.5 (v_sqrt_f32) and exp2(gamma * log2 x) otherwise -- about 12 instructions
// per element instead of ~60 for the correctly rounded powf / logf, which made the kernel
// VALU-bound (291 us vs 45 us of HBM time for the cfg-2 batch).
template <bool G15, bool GRAD>
__device__ __forceinline__ float focal_term(float pred, bool y, const LossParams lp, float &grad)
{
    const float eps = 1e-6f, ln2 = 0.69314718f;
    const float p = vmin(vmax(pred, eps), 1.0f - eps);                 // metric.py:22-23
    const float q = 1.0f - p;
    const float pt = y ? p : q;                                          // :28
    const float om = 1.0f - pt;
    const float alpha_t = y ? lp.alpha : 1.0f - lp.alpha;                // :25-26
    const float l2 = __builtin_amdgcn_logf(pt);                          // log2(pt)
    float w, lom = 0.0f;
    if (G15) {
        w = om * __builtin_amdgcn_sqrtf(om);
    } else {
        lom = __builtin_amdgcn_logf(om);
        w = __builtin_amdgcn_exp2f(lp.gamma * lom);
    }
    const float lg = ln2 * l2;
    if (GRAD) {
        // d/dpt [-a (1-pt)^g log pt] = a [ g (1-pt)^(g-1) log pt - (1-pt)^g / pt ];  dpt/dp = y ? 1 : -1;
        // clip_by_value passes the gradient only inside [eps, 1-eps]
        const float wm1 = G15 ? __builtin_amdgcn_sqrtf(om) : __builtin_amdgcn_exp2f((lp.gamma - 1.0f) * lom);
        const float dpt = alpha_t * (lp.gamma * wm1 * lg - w * __builtin_amdgcn_rcpf(pt));
        const bool inside = pred >= eps && pred <= 1.0f - eps;
        grad = inside ? (y ? dpt : -dpt) : 0.0f;
    }
    return (-alpha_t * w) * lg;                                          // :30
}

// The same element for y = 0 (pt = 1 - p), which is all but one element per foreground anchor: no
// label logic at all in the streaming loop; the matched class of a foreground anchor is corrected
// afterwards by that anchor's thread (match_loss_kernel).  cneg = -(1 - alpha) * ln2.
template <bool G15, bool GRAD>
__device__ __forceinline__ float focal_neg(float pred, const LossParams lp, float cneg, float &grad)
{
    const float eps = 1e-6f, ln2 = 0.69314718f;
    const float p = __builtin_amdgcn_fmed3f(pred, eps, 1.0f - eps);
    const float pt = 1.0f - p;
    const float om = 1.0f - pt;
    const float l2 = __builtin_amdgcn_logf(pt);
    float w, wm1, lom = 0.0f;
    if (G15) {
        wm1 = __builtin_amdgcn_sqrtf(om);
        w = om * wm1;
    } else {
        lom = __builtin_amdgcn_logf(om);
        w = __builtin_amdgcn_exp2f(lp.gamma * lom);
        wm1 = GRAD ? __builtin_amdgcn_exp2f((lp.gamma - 1.0f) * lom) : 0.0f;
    }
    if (GRAD) {
        // dL/dp = -dL/dpt = (1 - alpha) [ w / pt - gamma wm1 log pt ], zero where the clip is active
        const float gneg = (1.0f - lp.alpha) * (w * __builtin_amdgcn_rcpf(pt) - (lp.gamma * ln2) * wm1 * l2);
        grad = p == pred ? gneg : 0.0f;
    }
    return (cneg * w) * l2;
}

// Element type of the predictions and of their gradients: float, or _Float16 for the reference's
// --use_fp16 (mixed_float16) training, where metric.py:16-17 and Keras' Huber cast the fp16 model
// outputs to fp32 before any arithmetic.  Same here: widen on load (exact), fp32 arithmetic,
// gradients rounded to nearest even on store.  A 16-byte vector holds VEC elements.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));

// Device-resident factors of the backward entry point (see match_loss_kernel).
struct GradScale {
    const float *up_dist, *up_class, *normalizer;
    int scaled;
};

template <typename T> struct PredIO;
template <> struct PredIO<float> {
    static constexpr uint32_t VEC = 4;
    static constexpr int PREFETCH = LOSS_PREFETCH_F32;
    typedef v4f vec_t;
    typedef float4 box_t;
    static __device__ __forceinline__ vec_t fill(float x) { return vec_t{x, x, x, x}; }
    static __device__ __forceinline__ void unpack(vec_t v, float (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
    static __device__ __forceinline__ vec_t pack(const float (&g)[4]) { return vec_t{g[0], g[1], g[2], g[3]}; }
    static __device__ __forceinline__ float4 load_box(const box_t *p, size_t row) { return p[row]; }
    static __device__ __forceinline__ void store_box(box_t *p, size_t row, float4 v) { p[row] = v; }
};
template <> struct PredIO<_Float16> {
    static constexpr uint32_t VEC = 8;
    static constexpr int PREFETCH = LOSS_PREFETCH_F16;
    typedef v8h vec_t;
    typedef v4h box_t;
    static __device__ __forceinline__ vec_t fill(float x) { const _Float16 h = (_Float16)x; return vec_t{h, h, h, h, h, h, h, h}; }
    static __device__ __forceinline__ void unpack(vec_t v, float (&x)[8])
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
    }
    static __device__ __forceinline__ vec_t pack(const float (&g)[8])
    {
        vec_t v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (_Float16)g[k];
        return v;
    }
    static __device__ __forceinline__ float4 load_box(const box_t *p, size_t row)
    {
        const v4h v = p[row];
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    static __device__ __forceinline__ void store_box(box_t *p, size_t row, float4 v)
    {
        p[row] = v4h{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
};

// bfloat16 (the native mixed-precision format of this GPU; the reference itself only has fp16): a
// 16-bit storage type of our own, widened by a shift, rounded to nearest even on store (NaN kept).
struct bf16_t {
    uint16_t bits;
    __device__ __forceinline__ explicit operator float() const { return __uint_as_float((uint32_t)bits << 16); }
    __device__ __forceinline__ bf16_t() = default;
    __device__ __forceinline__ explicit bf16_t(float f)
    {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;            // NaN stays NaN
        else u += 0x7fffu + ((u >> 16) & 1u);                             // round to nearest even
        bits = (uint16_t)(u >> 16);
    }
};
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
// two bfloat16 per dword: the low one widens with a shift, the high one with a mask
__device__ __forceinline__ float bf16_lo(uint32_t d) { return __uint_as_float(d << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t d) { return __uint_as_float(d & 0xffff0000u); }
// gfx950 packs and rounds (to nearest even, NaN kept) two floats in one instruction
__device__ __forceinline__ uint32_t bf16_pair(float lo, float hi)
{
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <> struct PredIO<bf16_t> {
    static constexpr uint32_t VEC = 8;
    static constexpr int PREFETCH = LOSS_PREFETCH_F16;
    typedef v4u32 vec_t;
    typedef v2u32 box_t;
    static __device__ __forceinline__ vec_t fill(float x) { const uint32_t d = bf16_pair(x, x); return vec_t{d, d, d, d}; }
    static __device__ __forceinline__ void unpack(vec_t v, float (&x)[8])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[2 * k] = bf16_lo(v[k]);
            x[2 * k + 1] = bf16_hi(v[k]);
        }
    }
    static __device__ __forceinline__ vec_t pack(const float (&g)[8])
    {
        return vec_t{bf16_pair(g[0], g[1]), bf16_pair(g[2], g[3]), bf16_pair(g[4], g[5]), bf16_pair(g[6], g[7])};
    }
    static __device__ __forceinline__ float4 load_box(const box_t *p, size_t row)
    {
        const v2u32 v = p[row];
        return make_float4(bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1]));
    }
    static __device__ __forceinline__ void store_box(box_t *p, size_t row, float4 v)
    {
        p[row] = v2u32{bf16_pair(v.x, v.y), bf16_pair(v.z, v.w)};
    }
};

template <typename T, bool GRADS, bool G15, bool ALIGNED>
__global__ __launch_bounds__(ME_TILE) void match_loss_kernel(
    const float4 *__restrict__ anchors, const float4 *__restrict__ gt_boxes,
    const int32_t *__restrict__ gt_labels, int N, int G, int C, float img_h, float img_w,
    float neg_thr, float pos_thr, const typename PredIO<T>::box_t *__restrict__ pred_boxes,
    const T *__restrict__ pred_scores, LossParams lp, GradScale gsc,
    typename PredIO<T>::box_t *__restrict__ grad_boxes, T *__restrict__ grad_scores, float *__restrict__ partials,
    uint32_t tiles_per_image, uint32_t c_magic)
{
    using IO = PredIO<T>;
    constexpr uint32_t VEC = IO::VEC;
    constexpr int LOSS_PREFETCH = IO::PREFETCH;
    typedef typename IO::vec_t vec_t;
    __shared__ float s_live[ME_TILE];           // |x| = 1.0: anchor counts (state != -1), 0.0 = ignored; -1.0 = foreground
    __shared__ float s_red[4][ME_TILE / WAVE];
    __shared__ unsigned long long s_top[ME_TILE];   // per foreground anchor: (ordered top score, ~class), LDS atomic max

    // backward entry point: gradients times upstream / normaliser, all three read from device
    // memory here (NULL upstream = that loss is unused: zeros; NULL or zero normaliser = 1, like
    // metric.py:121), so no second pass over [B,N,C] and no host round trip for npos
    float gs_box = 1.0f, gs_cls = 1.0f;
    if (GRADS && gsc.scaled) {
        float nrm = gsc.normalizer ? *gsc.normalizer : 1.0f;
        nrm = nrm != 0.0f ? nrm : 1.0f;
        gs_box = (gsc.up_dist ? *gsc.up_dist : 0.0f) / nrm;
        gs_cls = (gsc.up_class ? *gsc.up_class : 0.0f) / nrm;
    }

    const uint32_t tid = threadIdx.x;
    // whole images per XCD, back to front like match_encode_kernel (same-box A/B on config 2: forward
    // 43.9 -> 42.7 us, fp16 forward 38.8 -> 37.4 us; the plain block order is slower with gradients)
    const uint32_t blk = gridDim.x - 1u - xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t b = blk / tiles_per_image;
    const uint32_t n0 = (blk - b * tiles_per_image) * ME_TILE;
    const uint32_t tile_n = min((uint32_t)ME_TILE, (uint32_t)N - n0);

    // The tile's [tile_n, C] predictions are one contiguous run: 16-byte loads over the flat element
    // range (scalar head / tail for the element-aligned ends; ALIGNED = C % VEC == 0 and a 16-byte
    // aligned tensor: no head, and a vector never straddles two anchors).  The first LOSS_PREFETCH
    // vectors of every thread are requested from inside match_tile, right after its own global
    // loads, so that part of the stream moves while the VALUs match (the match needs the memory
    // system for nothing else); the loop then keeps LOSS_PREFETCH loads in flight per thread.
    const uint32_t d = (uint32_t)C;
    const size_t soff = ((size_t)b * N + n0) * d;
    const T *p = pred_scores + soff;
    const uint32_t count = tile_n * d;
    // vectors start on a LOSS_ALIGN-byte line (scalar head elements before it), so a wave-wide access
    // covers whole cache lines; with ALIGNED the head is a multiple of VEC and rows stay unstraddled
    const uint32_t head = min(count, (uint32_t)(((LOSS_ALIGN - ((uintptr_t)p & (LOSS_ALIGN - 1u))) & (LOSS_ALIGN - 1u)) / sizeof(T)));
    const uint32_t nvec = (count - head) / VEC;
    const vec_t *pv4 = reinterpret_cast<const vec_t *>(p + head);
    vec_t pre[LOSS_PREFETCH];
#pragma unroll
    for (int k = 0; k < LOSS_PREFETCH; ++k) pre[k] = IO::fill(0.5f);
    auto prefetch = [&] {
        if (d >= VEC && nvec) {
#pragma unroll
            for (int k = 0; k < LOSS_PREFETCH; ++k) pre[k] = pv4[min(tid + (uint32_t)k * ME_TILE, nvec - 1)];
        }
    };

    const TileMatch tm = match_tile<ME_TILE, false>(anchors, gt_boxes, gt_labels, b, n0, N, G, C, img_h, img_w, neg_thr,
                                                    pos_thr, prefetch);
    s_live[tid] = tm.live && tm.state != -1 ? (tm.state == 1 ? -1.0f : 1.0f) : 0.0f;
    s_top[tid] = 0ull;
    // metric.py:128-130 (CategoricalAccuracy over the foreground anchors) rides on the stream: a
    // thread whose vector lies in a foreground row folds its elements into that row's (score, class)
    // maximum; forward launches only
    const bool want_hits = partials != nullptr;
    auto top_key = [](float x, uint32_t col) -> unsigned long long {
        uint32_t u = __float_as_uint(x + 0.0f);                      // -0 -> +0: equal scores must tie
        u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;                  // order-preserving bits
        return ((unsigned long long)u << 32) | (0xffffffffu - col);  // ties: the lower class wins (tf.argmax)
    };

    // the one y = 1 element of a foreground anchor: fetch it now, correct the sum after the stream
    const bool has_pos = tm.live && tm.state == 1 && tm.label >= 0;
    const uint32_t pos_e = tid * d + (uint32_t)max(tm.label, 0);
    const float pos_x = has_pos ? (float)p[pos_e] : 0.5f;

    // ---- box loss: Huber over the 4 offsets of every foreground anchor ------------------------
    float dist = 0.0f, npos = 0.0f;
    const size_t row = (size_t)b * N + n0 + tid;
    if (tm.live) {
        float4 gbox = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tm.state == 1) {
            const float4 a = tm.a;
            const float4 g = gt_boxes[(size_t)b * G + tm.best];
            const float aw = a.z - a.x, ah = a.w - a.y;
            const float t[4] = {(((g.x - a.x) / aw) - 0.0f) / 0.2f, (((g.y - a.y) / ah) - 0.0f) / 0.2f,
                                (((g.z - a.z) / aw) - 0.0f) / 0.2f, (((g.w - a.w) / ah) - 0.0f) / 0.2f};   // anchors.py:286-297
            const float4 pb = IO::load_box(pred_boxes, row);
            const float e[4] = {pb.x - t[0], pb.y - t[1], pb.z - t[2], pb.w - t[3]};
            float h = 0.0f, ge[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ae = fabsf(e[k]);
                const bool quad = ae <= lp.delta;
                h += quad ? 0.5f * (e[k] * e[k]) : lp.delta * ae - 0.5f * (lp.delta * lp.delta);
                ge[k] = ((quad ? e[k] : (e[k] > 0.0f ? lp.delta : -lp.delta)) * 0.25f) * gs_box;
            }
            dist = h * 0.25f;                                             // K.mean over the 4 coordinates
            npos = 1.0f;
            gbox = make_float4(ge[0], ge[1], ge[2], ge[3]);
        }
        if (GRADS && grad_boxes) IO::store_box(grad_boxes, row, gbox);
    }
    __syncthreads();

    // ---- class loss: every element as y = 0, masked by its anchor's live flag ------------------
    const float cneg = -(1.0f - lp.alpha) * 0.69314718f;
    float cls = 0.0f;
    T *gp = GRADS ? grad_scores + soff : nullptr;
    {
        auto one = [&](uint32_t e) {
            uint32_t q, r;
            divmod_u32(e, d, c_magic, q, r);
            const float lv = s_live[q];
            const float m = fabsf(lv);
            float g = 0.0f;
            const float x = (float)p[e];
            if (want_hits && lv < 0.0f) atomicMax(&s_top[q], top_key(x, r));
            cls += m * focal_neg<G15, GRADS>(x, lp, cneg, g);
            if (GRADS) gp[e] = (T)(g * (m * gs_cls));       // m is exactly 0 or 1
        };
        if (tid < head) one(tid);
        if (d >= VEC) {
            auto process = [&](uint32_t v, const vec_t cur) {
                const uint32_t e = head + VEC * v;
                uint32_t q, r;
                divmod_u32(e, d, c_magic, q, r);
                const float lv0 = s_live[q];
                const float m0 = fabsf(lv0);
                float in[VEC], g[VEC], l[VEC];
                IO::unpack(cur, in);
                if (ALIGNED && want_hits && lv0 < 0.0f) {
                    float bx = in[0];
                    uint32_t bk = 0;
#pragma unroll
                    for (uint32_t k = 1; k < VEC; ++k)
                        if (in[k] > bx) { bx = in[k]; bk = k; }
                    atomicMax(&s_top[q], top_key(bx, r + bk));
                }
#pragma unroll
                for (uint32_t k = 0; k < VEC; ++k) {
                    g[k] = 0.0f;
                    l[k] = focal_neg<G15, GRADS>(in[k], lp, cneg, g[k]);
                }
                if (ALIGNED) {
                    float s4 = (l[0] + l[1]) + (l[2] + l[3]);
                    if (VEC == 8) s4 += (l[VEC - 4] + l[VEC - 3]) + (l[VEC - 2] + l[VEC - 1]);
                    cls += m0 * s4;
#pragma unroll
                    for (uint32_t k = 0; k < VEC; ++k) g[k] *= m0 * gs_cls;
                } else {
                    // a 16-byte vector spans at most two anchors (d >= VEC): q (elements k < left) and q + 1
                    const uint32_t q1 = min(q + 1, (uint32_t)ME_TILE - 1);
                    const float lv1 = s_live[q1];
                    const float m1 = fabsf(lv1);
                    const uint32_t left = d - r;
                    if (want_hits && (lv0 < 0.0f || lv1 < 0.0f)) {
#pragma unroll
                        for (uint32_t k = 0; k < VEC; ++k) {
                            const bool first = k < left;
                            if ((first ? lv0 : lv1) < 0.0f) atomicMax(&s_top[first ? q : q1], top_key(in[k], first ? r + k : k - left));
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < VEC; ++k) {
                        const float m = k < left ? m0 : m1;
                        cls += m * l[k];
                        g[k] *= m * gs_cls;
                    }
                }
                if (GRADS) {
                    // streamed once, never re-read here: keep it out of the caches (77 vs 97 us per launch).
                    // A compiler-visible store on purpose: the matched-class fix-up below overwrites one of
                    // these elements after the barrier, and only stores the compiler knows about are
                    // waited for (s_waitcnt) in front of it -- an inline-asm store raced with the fix-up.
                    __builtin_nontemporal_store(IO::pack(g), reinterpret_cast<vec_t *>(gp + e));
                }
            };
            for (uint32_t v0 = tid; v0 < nvec; v0 += LOSS_PREFETCH * ME_TILE) {
#pragma unroll
                for (int k = 0; k < LOSS_PREFETCH; ++k) {
                    const uint32_t v = v0 + (uint32_t)k * ME_TILE;
                    const vec_t cur = pre[k];
                    pre[k] = pv4[min(v + LOSS_PREFETCH * ME_TILE, nvec - 1)];   // clamped => unconditional
                    if (v < nvec) process(v, cur);
                }
            }
        } else {
            for (uint32_t e = head + tid; e < head + VEC * nvec; e += ME_TILE) one(e);
        }
        const uint32_t done = head + VEC * nvec;
        if (tid < count - done) one(done + tid);
    }

    // ---- the matched class of every foreground anchor: replace its y = 0 term by the y = 1 term ---
    float hit = 0.0f;
    {
        float g0 = 0.0f, g1 = 0.0f;
        const float lneg = focal_neg<G15, GRADS>(pos_x, lp, cneg, g0);
        const float lpos = focal_term<G15, GRADS>(pos_x, true, lp, g1);
        if (has_pos) cls += lpos - lneg;
        if (GRADS || want_hits) __syncthreads();     // the streaming stores / LDS maxima of this tile are done
        if (GRADS && has_pos) gp[pos_e] = (T)(g1 * gs_cls);
        // an out-of-range label is an all-zero one-hot row, whose argmax is class 0
        if (want_hits && tm.live && tm.state == 1)
            hit = 0xffffffffu - (uint32_t)s_top[tid] == (uint32_t)max(tm.label, 0) ? 1.0f : 0.0f;
    }

    // ---- block partial sums (wave shuffle, then one thread adds the 4 wave sums in order) ------
    // (by DPP like the matcher's bounding box: quads, half rows, rows, row 0 into 1 and 2 into 3, 1 into 3 -- lane 63 holds
    // the wave's four sums; a fixed order, so the sums are the same bits run after run.  Six __shfl_down steps per sum were
    // 24 ds_bpermute in six dependent LDS round trips at the tail of every block.)
    float v3[4] = {cls, dist, npos, hit};
#define ODBOX_SUM_STEP(ctrl)                                        \
    "v_add_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n"              \
    "v_add_f32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n"              \
    "v_add_f32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n"              \
    "v_add_f32_dpp %3, %3, %3 " ctrl " bank_mask:0xf\n"
    asm volatile("s_nop 1\n"
                 ODBOX_SUM_STEP("quad_perm:[1,0,3,2] row_mask:0xf")
                 ODBOX_SUM_STEP("quad_perm:[2,3,0,1] row_mask:0xf")
                 ODBOX_SUM_STEP("row_half_mirror row_mask:0xf")
                 ODBOX_SUM_STEP("row_mirror row_mask:0xf")
                 ODBOX_SUM_STEP("row_bcast:15 row_mask:0xa")
                 ODBOX_SUM_STEP("row_bcast:31 row_mask:0xc")
                 "s_nop 1"
                 : "+v"(v3[0]), "+v"(v3[1]), "+v"(v3[2]), "+v"(v3[3]));
#undef ODBOX_SUM_STEP
    if ((tid & 63u) == 63u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_red[k][tid >> 6] = v3[k];
    }
    __syncthreads();
    if (tid < 4 && partials) {
        float x = 0.0f;
        for (int w = 0; w < ME_TILE / WAVE; ++w) x += s_red[tid][w];
        partials[(size_t)blk * 4 + tid] = x;
    }
}

// metric.py:9-38 focal_loss on dense targets (y_true == 1 marks the positive class), summed over
// all n elements; used for the whole-image classifier head (metric.py:132-134).  Per-block
// partials in slot 0 of the [nblocks, 4] buffer.
template <bool G15>
__global__ __launch_bounds__(256) void focal_dense_kernel(const float *__restrict__ y_true,
                                                          const float *__restrict__ y_pred, size_t n, LossParams lp,
                                                          float *__restrict__ grad, float *__restrict__ partials)
{
    __shared__ float s_red[4];
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        float g = 0.0f;
        acc += focal_term<G15, true>(y_pred[i], y_true[i] == 1.0f, lp, g);
        if (grad) grad[i] = g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63u) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * 4 + 0] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        partials[(size_t)blockIdx.x * 4 + 1] = 0.0f;
        partials[(size_t)blockIdx.x * 4 + 2] = 0.0f;
        partials[(size_t)blockIdx.x * 4 + 3] = 0.0f;
    }
}

// metric.py:9-38 row by row: the softmax branch (from_logits, :19-20: tf.nn.softmax over the last axis = exp(x - max) /
// sum) and the un-reduced return (:33-38: one sum over the class axis per row).  One wave per row; the row is re-read
// per pass out of L2 (this serves the whole-image head, [B, num_image_classes]: kilobytes).  grad (optional): d
// out_rows[row] / d y_pred -- through the softmax when from_logits: p_i (g_i - sum_j g_j p_j), g = the gradient w.r.t.
// the clipped probabilities.
template <bool G15>
__global__ __launch_bounds__(256) void focal_rows_kernel(const float *__restrict__ y_true, const float *__restrict__ y_pred,
                                                         uint32_t rows, uint32_t C, LossParams lp, int from_logits,
                                                         float *__restrict__ out_rows, float *__restrict__ grad)
{
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= rows) return;
    const float *yt = y_true + (size_t)row * C, *yp = y_pred + (size_t)row * C;
    auto wave_sum = [](float x) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        return x;
    };
    float mx = 0.0f, den = 1.0f;
    if (from_logits) {
        float m = -INFINITY;
        for (uint32_t c = lane; c < C; c += WAVE) m = fmaxf(m, yp[c]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float e = 0.0f;
        for (uint32_t c = lane; c < C; c += WAVE) e += expf(yp[c] - m);
        mx = m;
        den = wave_sum(e);
    }
    auto prob = [&](uint32_t c) { return from_logits ? expf(yp[c] - mx) / den : yp[c]; };
    float acc = 0.0f, dot = 0.0f;
    for (uint32_t c = lane; c < C; c += WAVE) {
        const float p = prob(c);
        float g = 0.0f;
        acc += focal_term<G15, true>(p, yt[c] == 1.0f, lp, g);
        dot += g * p;
    }
    acc = wave_sum(acc);
    dot = wave_sum(dot);
    if (lane == 0) out_rows[row] = acc;
    if (grad) {
        for (uint32_t c = lane; c < C; c += WAVE) {
            const float p = prob(c);
            float g = 0.0f;
            focal_term<G15, true>(p, yt[c] == 1.0f, lp, g);
            grad[(size_t)row * C + c] = from_logits ? p * (g - dot) : g;
        }
    }
}

// partials [nblocks, 4] (class_sum, dist_sum, npos, hits) -> out[0] = dist_loss, out[1] = class_loss
// (both divided by npos when npos != 0, metric.py:121-123), out[2] = npos, out[3] = hits (foreground
// anchors whose top predicted class is the matched one); sums in double, fixed order.
#ifndef ODBOX_RED_THREADS
#define ODBOX_RED_THREADS 512               // same-box A/B of the forward loss step: 256 -> 42.3 us, 512 -> 41.5, 1024 -> 42.3
#endif
constexpr int RED_THREADS = ODBOX_RED_THREADS;
__global__ __launch_bounds__(RED_THREADS) void loss_reduce_kernel(const float4 *__restrict__ partials, uint32_t nblocks,
                                                                  float *__restrict__ out)
{
    // One block; a single-block kernel is pure latency, so every thread issues its loads back to
    // back (4 rows per trip) before the first add.  Thread t owns rows t, t + 1024, ...: the
    // summation tree is fixed by (nblocks), not by scheduling.
    __shared__ double s_acc[4][RED_THREADS / WAVE];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (uint32_t i0 = threadIdx.x; i0 < nblocks; i0 += 4 * RED_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + (uint32_t)u * RED_THREADS;
            v[u] = i < nblocks ? partials[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] += (double)v[u].x;
            acc[1] += (double)v[u].y;
            acc[2] += (double)v[u].z;
            acc[3] += (double)v[u].w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
        if ((threadIdx.x & 63u) == 0) s_acc[k][threadIdx.x >> 6] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < RED_THREADS / WAVE; ++w)
            for (int k = 0; k < 4; ++k) t[k] += s_acc[k][w];
        const double cls = t[0], dist = t[1], npos = t[2];
        out[0] = (float)(npos != 0.0 ? dist / npos : dist);
        out[1] = (float)(npos != 0.0 ? cls / npos : cls);
        out[2] = (float)npos;
        out[3] = (float)t[3];
    }
}

// Per-GT best anchor (see match_tile): the keys start at 0 = "nobody posted" ...
__global__ __launch_bounds__(256) void gt_best_init_kernel(unsigned long long *__restrict__ keys, uint32_t total)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < total) keys[i] = 0ull;
}
// ... and are turned, in place, into the records of include/odbox.h: { int32 anchor; float iou }.
__global__ __launch_bounds__(256) void gt_best_finalize_kernel(const float4 *__restrict__ anchors,
                                                               const float4 *__restrict__ gt, int N, int G,
                                                               unsigned long long *keys, uint32_t total)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned long long k = keys[i];
    uint32_t n = k != 0ull ? 0xffffffffu - (uint32_t)k : 0u;        // nobody posted: every IoU is +0 (or -1), argmax = 0
    n = min(n, (uint32_t)N - 1u);
    const float4 a = anchors[n];
    const float4 gb = gt[i];                                         // i = b * G + g
    const float a_area = (a.w - a.y) * (a.z - a.x);
    const float g_area = (gb.w - gb.y) * (gb.z - gb.x);
    const float gmax = vmax(vmax(gb.x, gb.y), vmax(gb.z, gb.w));
    const float v = gmax < 0.0f ? -1.0f : pair_iou(a, a_area, gb, g_area);       // bndbox.py:289-297
    int2 rec;
    rec.x = (int)n;
    rec.y = __float_as_int(v);
    reinterpret_cast<int2 *>(keys)[i] = rec;
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
    float gmax = vmax(vmax(gb.x, gb.y), vmax(gb.z, gb.w));
    out[t] = gmax < 0.0f ? -1.0f : v;
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

// tuning hooks of this unit (ODBOX_SWEEP*): read only in a build made with ODBOX_TUNING_HOOKS=1, like odbox_infer.hip's
inline const char *train_tuning_env(const char *name)
{
#ifdef ODBOX_TUNING_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

int train_cu_count()
{
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached_dev = dev;
        cached_cus = cus;
    }
    return cached_cus;
}

// ---- ODBOX_POLICY_AUTO is a function of the call's shape, nothing else (round 3) ------------------------------
// Through the caches a launch is 1.2x faster when its stores land in lines the 256 MB Infinity Cache still holds (one
// buffer rewritten call after call with nothing in between: configs[1] 33.5 us against 40 us around the caches) and
// 1.25x SLOWER when they do not (a fresh or rotating buffer, or one that a training step's gigabytes have pushed out
// since: 49.6 against 39.5 us; same-box table in profiles/README.md, tools/exp_policy.py).  Whether the lines are still
// there depends on traffic the library cannot see.  Round 2 guessed from a host clock (5 ms) and a list of its own
// recent outputs; that made the kernel of a call depend on host jitter, froze a guess into captured HIP graphs and took
// a process-wide mutex per call.  Now AUTO assumes what holds in every real loop -- the buffer is cold -- and takes the
// sweep wherever it applies; a caller whose consumer reads the targets straight away, or who rewrites one buffer in a
// tight loop, says ODBOX_POLICY_CACHE.  No state, no clock, no lock: the same call always launches the same kernel, and
// odbox_match_encode_plan tells which.
thread_local int g_last_plan = -1;      // odbox_match_encode_last_plan

// does a streamed launch of odbox_match_encode take match_encode_sweep_kernel?
bool sweep_applies(uint64_t label_bytes, int B, int G, uint32_t d, bool labels_aligned16, bool gt_best)
{
    // (staging: every one of the 512 blocks looks at the whole batch's GT rows once, 20 bytes each; those reads -- out
    // of L2 after the first block -- must stay under half the bytes the launch writes, or the tiles are the better plan:
    // B = 8192, G = 64, N = 100 would read 5 GB to write 265 MB; up to 20 480 rows -- 400 KB per block -- nobody counts)
    bool sweep = label_bytes > 0 && !gt_best && labels_aligned16 && G <= SWEEP_MAX_G && B <= SWEEP_MAX_B &&
                 ((uint64_t)B * (uint64_t)G <= 20480ull || (uint64_t)B * (uint64_t)G * 20ull * 512ull * 2ull <= label_bytes) &&
                 (SWEEP_PIECE + d - 1u) / d + 1u <= (uint32_t)SWEEP_RPP &&
                 label_bytes / sizeof(float) <= 0xffffffffull - 4u * SWEEP_PIECE &&
                 label_bytes <= ((uint64_t)(train_tuning_env("ODBOX_SWEEP_MAX_MB") ? atoi(train_tuning_env("ODBOX_SWEEP_MAX_MB")) : ODBOX_SWEEP_MAX_MB) << 20);
    if (const char *e = train_tuning_env("ODBOX_SWEEP")) sweep = sweep && atoi(e) != 0;
    return sweep;
}

}  // namespace

extern "C" {

int odbox_version(void) { return ODBOX_VERSION; }

int odbox_set_output_policy(int policy)
{
    if (policy < ODBOX_POLICY_AUTO || policy > ODBOX_POLICY_STREAM) return fail(ODBOX_EINVAL, "odbox_set_output_policy: unknown policy %d", policy);
    return g_output_policy.exchange(policy, std::memory_order_relaxed);
}
const char *odbox_last_error(void) { return odbox::last_error(); }

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
                       float *out_max_iou, odbox_gt_best_t *out_gt_best, void *stream)
{
    if (B < 0 || N < 0 || G < 1 || C < 1 || C > ODBOX_MAX_CLASSES)
        return fail(ODBOX_EINVAL, "odbox_match_encode: bad shape B=%d N=%d G=%d C=%d (need G>=1, 1<=C<=%d)", B, N, G, C,
                    ODBOX_MAX_CLASSES);
    if (B == 0 || N == 0) return 0;
    if (!anchors || !gt_boxes) return fail(ODBOX_EINVAL, "odbox_match_encode: NULL input");
    if (out_labels && !gt_labels) return fail(ODBOX_EINVAL, "odbox_match_encode: out_labels needs gt_labels");
    if (!aligned16(anchors) || !aligned16(gt_boxes))
        return fail(ODBOX_EINVAL, "odbox_match_encode: anchors / gt_boxes must be 16-byte aligned");
    if ((uintptr_t)out_gt_best & 7u) return fail(ODBOX_EINVAL, "odbox_match_encode: out_gt_best must be 8-byte aligned");
    if (out_gt_best && (uint64_t)B * (uint64_t)G > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_match_encode: B * G too large");
    const uint32_t d = (uint32_t)C + 1u;
    // dense labels beyond twice the 256 MB Infinity Cache: bypass it, smaller tiles (see match_encode_kernel).
    // Tile order: whole images per XCD (xcd_remap) is balanced by construction and keeps an image's tiles
    // on one L2; it is the rule.  The plain order is 1-3 % faster for a launch that is bound by a store
    // stream the Infinity Cache absorbs (same-box sweep: wins up to 344 MB, loses from 413 MB), but it
    // maps tile t to XCD t % 8, which is badly unbalanced when the per-tile match cost matters and
    // tiles_per_image is a multiple of 8 (config 4: 622 vs 412 us).  So: plain order only when the match
    // is cheap (G <= 64: one culling step, hidden under the stores) and the labels are at most 384 MB.
    const uint64_t label_bytes = out_labels ? (uint64_t)B * (uint64_t)N * d * sizeof(float) : 0;
    // The caller may know better than the size rule (odbox_set_output_policy): outputs that nothing reads
    // soon (targets produced ahead of their consumer, a different buffer every call) should not go
    // through the caches even when they would fit -- configs[1] into cold buffers: 50.2 us through the
    // caches, 43.6 us around them; rewriting ONE buffer in place: 33.2 us against 43.6 us.
    const int policy = g_output_policy.load(std::memory_order_relaxed);
    // (AUTO's size rule stops at 80 GT rows: the streaming variant's one-wave blocks pay for long lists -- 1.03 GB of labels,
    // G = 20 / 65 / 100 / 200 real boxes: 176 / 217 / 255 / 373 us around the caches against 210 / 231 / 246 / 291 us through
    // them, tools/exp_policy_longgt.py 300)
    bool stream_out = label_bytes > 0 && (policy == ODBOX_POLICY_STREAM ||
                                          (policy == ODBOX_POLICY_AUTO && label_bytes > ((uint64_t)ODBOX_STREAM_MB << 20) &&
                                           G <= STREAM_AUTO_MAX_G));
    // Narrow label rows always go through the caches: the streaming variant's 64-anchor tiles shrink with the row (C = 20:
    // 5 KB per block) and its launch becomes block- and match-bound -- same-box, 229 MB of labels, cold / in place:
    // C = 20 111 / 111 us around the caches against 63 / 51 us through them, C = 40 62 / 62 against 55 / 36 us
    // (tools/exp_policy_narrow.py); from C + 1 = 64 floats per row the stream wins (C = 80: 43 against 49.5 us cold)
    stream_out = stream_out && d >= (uint32_t)STREAM_MIN_ROW;
    // AUTO below the size rule, for the shapes the sweep serves (a label stream with a cheap match): around the caches,
    // as for a buffer that is not in the Infinity Cache (see above: the library cannot know, and cold is the usual case)
    if (policy == ODBOX_POLICY_AUTO && !stream_out && sweep_applies(label_bytes, B, G, d, aligned16(out_labels), out_gt_best != nullptr))
        stream_out = true;
#ifdef ODBOX_ENC_REMAP_FORCE
    const uint32_t remap = ODBOX_ENC_REMAP_FORCE;
#else
    uint32_t remap = (!stream_out && label_bytes > 0 && label_bytes <= (384ull << 20) && G <= 64) ? 0u : 1u;
    // the matcher is the cost, not the stores: long lists first, chip-wide (XCD = image mod 8 once there are 8 images;
    // fewer would bring back the tile-mod-8 imbalance described above)
    if (!stream_out && G > 64 && B >= 8) remap = 2u;
    if (const char *e = train_tuning_env("ODBOX_ENC_REMAP")) remap = (uint32_t)atoi(e);
#endif
    const int tile = stream_out ? ME_TILE_STREAM : ME_TILE_ENC;
    const uint64_t tiles = ((uint64_t)N + tile - 1) / tile;
    const uint64_t blocks = tiles * (uint64_t)B;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_match_encode: grid too large");
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(out_gt_best);
    const uint32_t bg = (uint32_t)B * (uint32_t)G;
    // The streaming variant runs FIVE one-wave blocks per CU, not the eight its 19.7 KB of LDS would allow: unused dynamic
    // LDS pads every block to 28 160 B (22 granules of 1 280 B; six of them do not fit the CU's 160 KB).  A store stream
    // into cold memory is fastest when few tiles are being written at once: same-box A/B on the configs[2] share (2.26 GB
    // per launch), resident blocks per CU 10 / 8 / 6 / 5 / 4: 349 / 344 / 325-329 / 325-329 / 364 us in round 2; with the
    // round-3 matcher 6 / 5 / 4: 328-338 / 321-326 / 362-364 us on one box, 330 / 321 on another (5 wins by 1-3 %); the
    // configs[1] batch into cold buffers 43.4 -> 43.0 us; more blocks per CU (GT staging cut to 128 rows: 32 per CU) 45.3
    // and 360 us.
#ifndef ODBOX_ME_STREAM_BLOCK_LDS
#define ODBOX_ME_STREAM_BLOCK_LDS 28160
#endif
    auto stream_pad = [&](auto kernel, int which) -> size_t {
        if (!stream_out) return 0;
        // static LDS of the instantiation, asked once each (it moves with ME_GCHUNK and the matcher's staging layout;
        // the two streaming instantiations share a signature, hence the index)
        static std::atomic<size_t> fixed_lds[2];
        size_t fixed = fixed_lds[which].load(std::memory_order_relaxed);
        if (fixed == 0) {
            hipFuncAttributes at{};
            fixed = hipFuncGetAttributes(&at, reinterpret_cast<const void *>(kernel)) == hipSuccess && at.sharedSizeBytes
                        ? (size_t)at.sharedSizeBytes : (size_t)ODBOX_ME_STREAM_BLOCK_LDS;
            fixed_lds[which].store(fixed, std::memory_order_relaxed);
        }
        return fixed < (size_t)ODBOX_ME_STREAM_BLOCK_LDS ? (size_t)ODBOX_ME_STREAM_BLOCK_LDS - fixed : 0;
    };
    auto launch = [&](auto kernel) {
        const size_t pad_lds = stream_pad(kernel, 0);
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(tile), pad_lds, (hipStream_t)stream,
                           (const float4 *)anchors, (const float4 *)gt_boxes, gt_labels, N, G, C, img_h, img_w,
                           neg_thr, pos_thr, out_regression, out_labels, out_match_idx, out_state, out_max_iou,
                           keys, (uint32_t)tiles, remap);
    };
    // streamed wide-row labels with a short GT list: the lockstep sweep (match_encode_sweep_kernel)
    const uint64_t total_floats = label_bytes / sizeof(float);
    const bool sweep = stream_out && sweep_applies(label_bytes, B, G, d, aligned16(out_labels), keys != nullptr);
    if (sweep) {
        const uint32_t pieces = (uint32_t)((total_floats + SWEEP_PIECE - 1) / SWEEP_PIECE);
        uint32_t grid = 2u * (uint32_t)train_cu_count();
        if (const char *e = train_tuning_env("ODBOX_SWEEP_GRID")) grid = atoi(e) > 0 ? (uint32_t)atoi(e) : grid;
        if (grid > pieces) grid = pieces;
        // rows of every image kept in LDS: 20 bytes each, beside one 16-bit bound per image, in SWEEP_LDS_ROWS * 20 bytes
        uint32_t gs = (uint32_t)SWEEP_LDS_ROWS * 20u > 2u * (uint32_t)B ? ((uint32_t)SWEEP_LDS_ROWS * 20u - 2u * (uint32_t)B) / (20u * (uint32_t)B) : 0u;
        if (gs > (uint32_t)G) gs = (uint32_t)G;
        if (const char *e = train_tuning_env("ODBOX_SWEEP_GT_LDS")) gs = atoi(e) != 0 ? gs : 0u;
        const size_t lds = ((size_t)B * gs * 20 + (size_t)B * 2 + 15) & ~(size_t)15;
        uint32_t xcd_runs = (grid & 7u) == 0u && label_bytes <= (384ull << 20) ? 1u : 0u;      // see the kernel
        if (const char *e = train_tuning_env("ODBOX_SWEEP_XCD_RUNS")) xcd_runs = xcd_runs && atoi(e) != 0;
        hipLaunchKernelGGL(match_encode_sweep_kernel, dim3(grid), dim3(WAVE * (SWEEP_SW + SWEEP_MW)), lds, (hipStream_t)stream,
                           (const float4 *)anchors, (const float4 *)gt_boxes, gt_labels, N, G, C, img_h, img_w, neg_thr,
                           pos_thr, out_regression, out_labels, out_match_idx, out_state, out_max_iou,
                           (uint32_t)((uint64_t)B * N), (uint32_t)total_floats, pieces, (uint32_t)B, gs, xcd_runs);
        g_last_plan = ODBOX_PLAN_SWEEP;
        return launch_status("match_encode_sweep");
    }
    g_last_plan = stream_out ? ODBOX_PLAN_TILES_STREAM : ODBOX_PLAN_TILES;
    if (!keys) {
        if (stream_out) launch(match_encode_kernel<true, false>);
        else launch(match_encode_kernel<false, false>);
        return launch_status("match_encode");
    }
    // per-GT best anchor: the records double as the 64-bit keys while the matcher runs
    hipLaunchKernelGGL(gt_best_init_kernel, dim3((bg + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, keys, bg);
    if (int rc = launch_status("gt_best_init")) return rc;
    // tile order of the per-GT variant: multiples of a number near 0.618 * tiles that is coprime to it (see the kernel)
    uint32_t mult = (uint32_t)((double)tiles * 0.6180339887) | 1u;
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(mult, tiles) != 1) mult += 2;
    if (train_tuning_env("ODBOX_GTBEST_LINEAR")) mult = 1;
    if (const char *e = train_tuning_env("ODBOX_GTBEST_MULT")) mult = (uint32_t)atoi(e);        // (must be coprime to the tile count)
    auto launch_best = [&](auto kernel) {
        const size_t pad_lds = stream_pad(kernel, 1);
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(tile), pad_lds, (hipStream_t)stream,
                           (const float4 *)anchors, (const float4 *)gt_boxes, gt_labels, N, G, C, img_h, img_w,
                           neg_thr, pos_thr, out_regression, out_labels, out_match_idx, out_state, out_max_iou,
                           keys, (uint32_t)tiles, mult);
    };
    if (stream_out) launch_best(match_encode_kernel<true, true>);
    else launch_best(match_encode_kernel<false, true>);
    if (int rc = launch_status("match_encode")) return rc;
    hipLaunchKernelGGL(gt_best_finalize_kernel, dim3((bg + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)anchors, (const float4 *)gt_boxes, N, G, keys, bg);
    return launch_status("gt_best_finalize");
}

int odbox_match_encode_plan(int B, int N, int G, int C, int labels_aligned16, int want_gt_best)
{
    if (B < 1 || N < 1 || G < 1 || C < 1 || C > ODBOX_MAX_CLASSES)
        return fail(ODBOX_EINVAL, "odbox_match_encode_plan: bad shape B=%d N=%d G=%d C=%d", B, N, G, C);
    const uint32_t d = (uint32_t)C + 1u;
    const uint64_t label_bytes = (uint64_t)B * (uint64_t)N * d * sizeof(float);
    const int policy = g_output_policy.load(std::memory_order_relaxed);
    const bool stream_out = (policy == ODBOX_POLICY_STREAM ||
                             (policy == ODBOX_POLICY_AUTO && label_bytes > ((uint64_t)ODBOX_STREAM_MB << 20) && G <= STREAM_AUTO_MAX_G)) &&
                            d >= (uint32_t)STREAM_MIN_ROW;
    const bool sweep = sweep_applies(label_bytes, B, G, d, labels_aligned16 != 0, want_gt_best != 0);
    if (policy == ODBOX_POLICY_AUTO && !stream_out) return sweep ? ODBOX_PLAN_SWEEP : ODBOX_PLAN_TILES;      // a buffer not written a moment ago
    if (!stream_out) return ODBOX_PLAN_TILES;
    return sweep ? ODBOX_PLAN_SWEEP : ODBOX_PLAN_TILES_STREAM;
}

int odbox_match_encode_last_plan(void) { return g_last_plan; }

#ifdef ODBOX_COUNT_PAIRS
int odbox_debug_pair_count(unsigned long long *out4, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_pair_count), sizeof(g_pair_count)));
    if (reset) {
        static const unsigned long long zero[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_count), zero, sizeof(zero)));
    }
    return 0;
}
#endif

size_t odbox_match_loss_workspace(int B, int N)
{
    if (B < 1 || N < 1) return 0;
    const size_t tiles = ((size_t)N + ME_TILE - 1) / ME_TILE;
    return (tiles * (size_t)B * 4 * sizeof(float) + 255) / 256 * 256;
}

// Shared by odbox_match_loss (out_loss + optional unscaled gradients) and odbox_match_loss_backward
// (gradients only, times upstream / normaliser read on the device).
// T = element type of the four prediction / gradient tensors (float or _Float16).
extern "C++" {
template <typename T>
static int match_loss_run(const char *name, const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B,
                          int N, int G, int C, float img_h, float img_w, float neg_thr, float pos_thr,
                          const T *pred_boxes, const T *pred_scores, float gamma, float alpha, float huber_delta,
                          float *out_loss, GradScale gsc, T *grad_boxes, T *grad_scores,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    typedef typename PredIO<T>::box_t box_t;
    constexpr uintptr_t box_mask = sizeof(box_t) - 1, el_mask = sizeof(T) - 1;
    if (B < 1 || N < 1 || G < 1 || C < 1 || C > ODBOX_MAX_CLASSES)
        return fail(ODBOX_EINVAL, "%s: bad shape B=%d N=%d G=%d C=%d (need B,N,G>=1, 1<=C<=%d)", name, B, N, G, C,
                    ODBOX_MAX_CLASSES);
    if (!anchors || !gt_boxes || !gt_labels || !pred_boxes || !pred_scores) return fail(ODBOX_EINVAL, "%s: NULL pointer", name);
    if ((grad_boxes == nullptr) != (grad_scores == nullptr))
        return fail(ODBOX_EINVAL, "%s: pass both gradient buffers or neither", name);
    if (!aligned16(anchors) || !aligned16(gt_boxes) || ((uintptr_t)pred_boxes & box_mask) || ((uintptr_t)grad_boxes & box_mask))
        return fail(ODBOX_EINVAL, "%s: anchors / gt_boxes must be 16-byte aligned, pred_boxes / grad_boxes %zu-byte", name,
                    sizeof(box_t));
    if (((uintptr_t)pred_scores & el_mask) || (grad_scores && ((uintptr_t)pred_scores & 15u) != ((uintptr_t)grad_scores & 15u)))
        return fail(ODBOX_EINVAL, "%s: grad_scores must have the 16-byte phase of pred_scores", name);
    if ((((uintptr_t)gsc.up_dist | (uintptr_t)gsc.up_class | (uintptr_t)gsc.normalizer) & 3u))
        return fail(ODBOX_EINVAL, "%s: upstream / normalizer scalars must be 4-byte aligned", name);
    if (out_loss) {
        const size_t need = odbox_match_loss_workspace(B, N);
        if (!workspace || workspace_bytes < need)
            return fail(ODBOX_ENOSPC, "%s: workspace %zu bytes, need %zu", name, workspace_bytes, need);
        if (!aligned16(workspace)) return fail(ODBOX_EINVAL, "%s: workspace must be 16-byte aligned", name);
    }
    const uint64_t tiles = ((uint64_t)N + ME_TILE - 1) / ME_TILE;
    const uint64_t blocks = tiles * (uint64_t)B;
    if (blocks > 0x7fffffffull) return fail(ODBOX_EINVAL, "%s: grid too large", name);
    const uint32_t d = (uint32_t)C;
    const uint32_t magic = d == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / d);
    const LossParams lp = {gamma, alpha, huber_delta};
    hipStream_t st = (hipStream_t)stream;
    const bool g15 = gamma == 1.5f;                 // the reference's default: pow as x * sqrt(x)
    const bool al = d % PredIO<T>::VEC == 0 && aligned16(pred_scores);
    float *partials = out_loss ? (float *)workspace : nullptr;
    // (tuning builds: ODBOX_LOSS_PAD bytes of unused dynamic LDS = fewer resident blocks per CU; see profiles/README.md)
    const char *pad_env = train_tuning_env("ODBOX_LOSS_PAD");
    const size_t pad_lds = pad_env ? (size_t)atoi(pad_env) : 0;
    auto launch = [&](auto kernel, box_t *gb, T *gs) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(ME_TILE), pad_lds, st, (const float4 *)anchors,
                           (const float4 *)gt_boxes, gt_labels, N, G, C, img_h, img_w, neg_thr, pos_thr,
                           (const box_t *)pred_boxes, pred_scores, lp, gsc, gb, gs, partials, (uint32_t)tiles,
                           magic);
    };
    auto pick = [&](auto grads, box_t *gb, T *gs) {
        constexpr bool GR = decltype(grads)::value;
        if (g15) { if (al) launch(match_loss_kernel<T, GR, true, true>, gb, gs); else launch(match_loss_kernel<T, GR, true, false>, gb, gs); }
        else     { if (al) launch(match_loss_kernel<T, GR, false, true>, gb, gs); else launch(match_loss_kernel<T, GR, false, false>, gb, gs); }
    };
    if (grad_scores) pick(std::true_type{}, (box_t *)grad_boxes, grad_scores);
    else pick(std::false_type{}, (box_t *)nullptr, (T *)nullptr);
    if (int rc = launch_status("match_loss")) return rc;
    if (!out_loss) return 0;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(RED_THREADS), 0, st, (const float4 *)workspace, (uint32_t)blocks, out_loss);
    return launch_status("loss_reduce");
}
}  // extern "C++"

int odbox_match_loss(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N, int G,
                     int C, float img_h, float img_w, float neg_thr, float pos_thr, const float *pred_boxes,
                     const float *pred_scores, float gamma, float alpha, float huber_delta, float *out_loss,
                     float *grad_boxes, float *grad_scores, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!out_loss) return fail(ODBOX_EINVAL, "odbox_match_loss: NULL pointer");
    return match_loss_run<float>("odbox_match_loss", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w, neg_thr, pos_thr,
                          pred_boxes, pred_scores, gamma, alpha, huber_delta, out_loss, GradScale{nullptr, nullptr, nullptr, 0},
                          grad_boxes, grad_scores, workspace, workspace_bytes, stream);
}

int odbox_match_loss_backward(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N, int G,
                              int C, float img_h, float img_w, float neg_thr, float pos_thr, const float *pred_boxes,
                              const float *pred_scores, float gamma, float alpha, float huber_delta,
                              const float *upstream_dist, const float *upstream_class, const float *normalizer,
                              float *grad_boxes, float *grad_scores, void *stream)
{
    if (!grad_boxes || !grad_scores) return fail(ODBOX_EINVAL, "odbox_match_loss_backward: NULL pointer");
    return match_loss_run<float>("odbox_match_loss_backward", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w, neg_thr,
                          pos_thr, pred_boxes, pred_scores, gamma, alpha, huber_delta, nullptr,
                          GradScale{upstream_dist, upstream_class, normalizer, 1}, grad_boxes, grad_scores, nullptr, 0, stream);
}

int odbox_match_loss_f16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N, int G,
                         int C, float img_h, float img_w, float neg_thr, float pos_thr, const void *pred_boxes,
                         const void *pred_scores, float gamma, float alpha, float huber_delta, float *out_loss,
                         void *grad_boxes, void *grad_scores, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!out_loss) return fail(ODBOX_EINVAL, "odbox_match_loss_f16: NULL pointer");
    return match_loss_run<_Float16>("odbox_match_loss_f16", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w, neg_thr,
                                    pos_thr, (const _Float16 *)pred_boxes, (const _Float16 *)pred_scores, gamma, alpha,
                                    huber_delta, out_loss, GradScale{nullptr, nullptr, nullptr, 0}, (_Float16 *)grad_boxes,
                                    (_Float16 *)grad_scores, workspace, workspace_bytes, stream);
}

int odbox_match_loss_backward_f16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N,
                                  int G, int C, float img_h, float img_w, float neg_thr, float pos_thr,
                                  const void *pred_boxes, const void *pred_scores, float gamma, float alpha,
                                  float huber_delta, const float *upstream_dist, const float *upstream_class,
                                  const float *normalizer, void *grad_boxes, void *grad_scores, void *stream)
{
    if (!grad_boxes || !grad_scores) return fail(ODBOX_EINVAL, "odbox_match_loss_backward_f16: NULL pointer");
    return match_loss_run<_Float16>("odbox_match_loss_backward_f16", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w,
                                    neg_thr, pos_thr, (const _Float16 *)pred_boxes, (const _Float16 *)pred_scores, gamma,
                                    alpha, huber_delta, nullptr, GradScale{upstream_dist, upstream_class, normalizer, 1},
                                    (_Float16 *)grad_boxes, (_Float16 *)grad_scores, nullptr, 0, stream);
}

int odbox_match_loss_bf16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N, int G,
                          int C, float img_h, float img_w, float neg_thr, float pos_thr, const void *pred_boxes,
                          const void *pred_scores, float gamma, float alpha, float huber_delta, float *out_loss,
                          void *grad_boxes, void *grad_scores, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!out_loss) return fail(ODBOX_EINVAL, "odbox_match_loss_bf16: NULL pointer");
    return match_loss_run<bf16_t>("odbox_match_loss_bf16", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w, neg_thr,
                                  pos_thr, (const bf16_t *)pred_boxes, (const bf16_t *)pred_scores, gamma, alpha,
                                  huber_delta, out_loss, GradScale{nullptr, nullptr, nullptr, 0}, (bf16_t *)grad_boxes,
                                  (bf16_t *)grad_scores, workspace, workspace_bytes, stream);
}

int odbox_match_loss_backward_bf16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N,
                                   int G, int C, float img_h, float img_w, float neg_thr, float pos_thr,
                                   const void *pred_boxes, const void *pred_scores, float gamma, float alpha,
                                   float huber_delta, const float *upstream_dist, const float *upstream_class,
                                   const float *normalizer, void *grad_boxes, void *grad_scores, void *stream)
{
    if (!grad_boxes || !grad_scores) return fail(ODBOX_EINVAL, "odbox_match_loss_backward_bf16: NULL pointer");
    return match_loss_run<bf16_t>("odbox_match_loss_backward_bf16", anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w,
                                  neg_thr, pos_thr, (const bf16_t *)pred_boxes, (const bf16_t *)pred_scores, gamma,
                                  alpha, huber_delta, nullptr, GradScale{upstream_dist, upstream_class, normalizer, 1},
                                  (bf16_t *)grad_boxes, (bf16_t *)grad_scores, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------
// The allocation-free variant of odbox_match_loss (no counterpart in the reference): for callers
// that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc.  The one scratch buffer of the loss path (the per-tile partial sums that
// loss_reduce_kernel adds up in a fixed order) lives in this library's own device memory: every
// GPU the library runs on holds its own copy of `scratch` until the process ends.
//   - `scratch` is a host-side stand-in in host code, never a device address: its address on the
//     current device comes from hipGetSymbolAddress, once per device, and only that pointer is
//     handed to kernels;
//   - nothing in it has to start at zero: every tile of a call writes its own 16 bytes before the
//     reducer (next launch on the same stream) reads exactly those; what a previous, larger call
//     left behind the last tile is never read;
//   - every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams,
//     two host threads) race.  Calls on one stream, or ordered by events, are fine.
// ------------------------------------------------------------------------------------------
constexpr uint32_t NOALLOC_MAX_TILES = ODBOX_NOALLOC_MAX_TILES;             // B * ceil(N / 256) of one call; 1 MiB per GPU
static __device__ __align__(256) float scratch[NOALLOC_MAX_TILES * 4];

extern "C++" {
namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
float *g_scratch_ptr[MAX_DEVICES];                        // device address of `scratch`, per GPU

// The current device's copy of `scratch` (set-up on the first call on that device).
// The kernels run on the STREAM's device with the address looked up here, so the two must be the same GPU on every
// call (another GPU's copy would be a peer access at best, and two GPUs sharing one array race).
int scratch_on_current_device(const char *name, void *stream, float **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(ODBOX_EINVAL, "%s: device ordinal %d out of range", name, dev);
    if (stream) {
        hipDevice_t sdev = dev;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && (int)sdev != dev)
            return fail(ODBOX_EINVAL, "%s: the stream belongs to device %d but the current device is %d: make the stream's "
                        "device current for the call", name, (int)sdev, dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float *p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(ODBOX_EHIP, "%s: hipGetSymbolAddress returned NULL", name);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace
}  // extern "C++"

int odbox_match_loss_noalloc(const float *anchors, const float *gt_boxes, const int32_t *gt_labels, int B, int N, int G,
                             int C, float img_h, float img_w, float neg_thr, float pos_thr, const float *pred_boxes,
                             const float *pred_scores, float gamma, float alpha, float huber_delta, float *out_loss,
                             float *grad_boxes, float *grad_scores, void *stream)
{
    const char *name = "odbox_match_loss_noalloc";
    if (!out_loss) return fail(ODBOX_EINVAL, "%s: NULL pointer", name);
    if (B < 1 || N < 1) return fail(ODBOX_EINVAL, "%s: bad shape B=%d N=%d (need B,N>=1)", name, B, N);
    const uint64_t tiles = (((uint64_t)N + ME_TILE - 1) / ME_TILE) * (uint64_t)B;
    if (tiles > NOALLOC_MAX_TILES)
        return fail(ODBOX_ENOSPC, "%s: B * ceil(N / %d) = %llu tiles, the library-owned scratch holds %u; use "
                    "odbox_match_loss with a caller workspace", name, ME_TILE, (unsigned long long)tiles, NOALLOC_MAX_TILES);
    float *ws = nullptr;
    if (int rc = scratch_on_current_device(name, stream, &ws)) return rc;
    return match_loss_run<float>(name, anchors, gt_boxes, gt_labels, B, N, G, C, img_h, img_w, neg_thr, pos_thr, pred_boxes,
                                 pred_scores, gamma, alpha, huber_delta, out_loss, GradScale{nullptr, nullptr, nullptr, 0},
                                 grad_boxes, grad_scores, ws, sizeof(float) * 4 * NOALLOC_MAX_TILES, stream);
}

size_t odbox_focal_loss_workspace(size_t n)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (blocks * 4 * sizeof(float) + 255) / 256 * 256;
}

int odbox_focal_loss(const float *y_true, const float *y_pred, size_t n, float gamma, float alpha, float *out_sum,
                     float *grad, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!out_sum) return fail(ODBOX_EINVAL, "odbox_focal_loss: NULL out_sum");
    if (n && (!y_true || !y_pred)) return fail(ODBOX_EINVAL, "odbox_focal_loss: NULL input");
    const size_t need = odbox_focal_loss_workspace(n);
    if (!workspace || workspace_bytes < need)
        return fail(ODBOX_ENOSPC, "odbox_focal_loss: workspace %zu bytes, need %zu", workspace_bytes, need);
    if (!aligned16(workspace)) return fail(ODBOX_EINVAL, "odbox_focal_loss: workspace must be 16-byte aligned");
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const LossParams lp = {gamma, alpha, 1.0f};
    hipStream_t st = (hipStream_t)stream;
    if (gamma == 1.5f)
        hipLaunchKernelGGL(focal_dense_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, st, y_true, y_pred, n, lp, grad,
                           (float *)workspace);
    else
        hipLaunchKernelGGL(focal_dense_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, st, y_true, y_pred, n, lp, grad,
                           (float *)workspace);
    if (int rc = launch_status("focal_dense")) return rc;
    // reuse the 4-slot reducer: slot 0 holds the sum, npos slot is 0 so nothing is divided; the
    // sum comes out in out[1]; out_sum must therefore have room for 4 floats
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(RED_THREADS), 0, st, (const float4 *)workspace, (uint32_t)blocks, out_sum);
    return launch_status("loss_reduce");
}

int odbox_focal_loss_rows(const float *y_true, const float *y_pred, size_t rows, int C, float gamma, float alpha,
                          int from_logits, float *out_rows, float *grad, void *stream)
{
    if (C < 1) return fail(ODBOX_EINVAL, "odbox_focal_loss_rows: C=%d", C);
    if (rows == 0) return 0;
    if (!y_true || !y_pred || !out_rows) return fail(ODBOX_EINVAL, "odbox_focal_loss_rows: NULL pointer");
    if (rows > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_focal_loss_rows: too many rows");
    const LossParams lp = {gamma, alpha, 1.0f};
    const dim3 grid((uint32_t)((rows + 3) / 4));
    if (gamma == 1.5f)
        hipLaunchKernelGGL(focal_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, y_true, y_pred, (uint32_t)rows,
                           (uint32_t)C, lp, from_logits, out_rows, grad);
    else
        hipLaunchKernelGGL(focal_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y_true, y_pred, (uint32_t)rows,
                           (uint32_t)C, lp, from_logits, out_rows, grad);
    return launch_status("focal_rows");
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

}  // extern "C"
