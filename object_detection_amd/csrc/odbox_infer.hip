// odbox_infer.hip -- inference-side kernels (score scan, candidate partition / sort, per-class
// greedy NMS, final top-k) + odbox_decode_nms (include/odbox.h), gfx950 / wave64.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-honor-nans (see build.py).
// No FMA contraction (each fp32 op rounds separately, like the reference's TF ops) and IEEE
// division.  -fno-honor-nans changes no rounding; scores and boxes are finite by contract and
// it lets v_min/v_max be emitted without a canonicalising pre-op on every operand.
//
// Reference citations are file:line under the reference tree (bioothod/object_detection).
#include "odbox_common.h"
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

using namespace odbox;

namespace {

// ------------------------------------------------------------------------------------------
// decode + NMS, kernel 1: stream the [B,N,C] scores once (the HBM-bound kernel).
//
// Persistent single-wave workgroups, each with two LDS buffers.  A tile = R consecutive rows
// of the score matrix = one contiguous run of R*C floats.  Tile i+1 goes HBM -> LDS by LDS-DMA
// (global_load_lds, 16 B per lane when the run is 16-byte aligned, else 4 B; no VGPR staging)
// while tile i is reduced out of the other buffer.  In the reduce, LPR adjacent lanes share a
// row and read V*LPR consecutive floats per instruction (ds_read_b32/b64/b128 by C's
// divisibility), so LDS reads are bank-conflict free for every C without padding (the DMA image
// must stay linear); the row maximum is finished with xor-shuffles inside the lane group.
// Only rows with max > threshold (rare) take the second pass that finds tf.argmax's FIRST
// maximal column, and append one 64-bit key to the image's candidate list (one atomic per
// wave-step).  bndbox.py:153-154 (argmax / reduce_max), :205 (score filter).
// ------------------------------------------------------------------------------------------
struct ScanGeom {
    int V;             // floats per LDS read: 4 if C%4==0, 2 if C%2==0, else 1
    int lpr_log2;      // log2(lanes per row)
    int K;             // reads per lane per row = ceil(C / (V * LPR))
    int kfull;         // leading reads that are in range for every lane = C / (V * LPR)
    int steps;         // wave-steps per tile; rows per tile R = steps * (64 >> lpr_log2)
    uint32_t buf_floats;   // floats per LDS buffer (multiple of 4)
    uint32_t stream;       // 1: nt loads (tensor far beyond the Infinity Cache)
    uint32_t tpi_magic;    // floor(2^32 / tiles_per_image) for the tile -> image division
};

// LDS-DMA issued through inline asm on purpose: with the builtin, hipcc treats every later
// ds_read as dependent on every outstanding DMA and puts s_waitcnt vmcnt(0) in front of it,
// which would serialise the prefetch of tile i+1 with the reduce of tile i.  Here the waits are
// ours: one s_waitcnt vmcnt(0) before a buffer is read.  M0 = wave-uniform LDS byte address;
// lane l lands at M0 + l * size.  M0 is written and consumed inside ONE asm statement, so nothing
// is scheduled between the two instructions.  M0 is a reserved register (hipcc rejects it in a
// clobber list), so that the compiler keeps no value of its own in it across the statement is
// CHECKED instead: build.py disassembles this unit and fails the build if any instruction other
// than these s_mov_b32 names m0 (check_m0).  s_nop covers the SALU-writes-M0 -> LDS-DMA hazard.
// `stream` (wave-uniform): the `nt` cache policy -- a score tensor far beyond the 256 MB Infinity
// Cache is read once and should not be kept anywhere (config 5: 192 -> 176 us); a small one may
// still sit in that cache from its producer, where nt loads were measured slower (loss kernel).
__device__ __forceinline__ void dma16(const float *g, uint32_t lds_byte, bool stream)
{
    if (stream)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(lds_byte), "v"(g) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g) : "memory");
}
__device__ __forceinline__ void dma4(const float *g, uint32_t lds_byte, bool stream)
{
    if (stream)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt" ::"s"(lds_byte), "v"(g) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_byte), "v"(g) : "memory");
}
__device__ __forceinline__ uint32_t lds_address(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// contiguous run of nfloat floats: HBM -> LDS image at byte address `lds`, linear
template <int VEC>
__device__ __forceinline__ void dma_tile(const float *src, uint32_t lds, uint32_t nfloat, uint32_t lane, bool stream)
{
    if constexpr (VEC == 4) {
        const uint32_t nvec = nfloat >> 2;
        uint32_t i = 0;
        for (; i + WAVE <= nvec; i += WAVE) dma16(src + (size_t)(i + lane) * 4, lds + i * 16, stream);
        if (i + lane < nvec) dma16(src + (size_t)(i + lane) * 4, lds + i * 16, stream);
        const uint32_t done = nvec << 2;
        if (done + lane < nfloat) dma4(src + done + lane, lds + done * 4, stream);
    } else {
        uint32_t i = 0;
        for (; i + WAVE <= nfloat; i += WAVE) dma4(src + i + lane, lds + i * 4, stream);
        if (i + lane < nfloat) dma4(src + i + lane, lds + i * 4, stream);
    }
}

// all-reduce inside aligned groups of 2^lpr_log2 lanes.  Up to 16 lanes it is pure DPP (quad
// permutes, row_half_mirror, row_mirror): VALU-rate, no trip through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int LL>
__device__ __forceinline__ float group_max(float m)
{
    if constexpr (LL >= 1) m = fmaxf(m, __int_as_float(dpp_i<0xB1>(__float_as_int(m))));    // quad_perm [1,0,3,2]
    if constexpr (LL >= 2) m = fmaxf(m, __int_as_float(dpp_i<0x4E>(__float_as_int(m))));    // quad_perm [2,3,0,1]
    if constexpr (LL >= 3) m = fmaxf(m, __int_as_float(dpp_i<0x141>(__float_as_int(m))));   // row_half_mirror
    if constexpr (LL >= 4) m = fmaxf(m, __int_as_float(dpp_i<0x140>(__float_as_int(m))));   // row_mirror
    if constexpr (LL >= 5) m = fmaxf(m, __shfl_xor(m, 16));
    if constexpr (LL >= 6) m = fmaxf(m, __shfl_xor(m, 32));
    return m;
}
template <int LL>
__device__ __forceinline__ uint32_t group_min(uint32_t m)
{
    if constexpr (LL >= 1) m = min(m, (uint32_t)dpp_i<0xB1>((int)m));
    if constexpr (LL >= 2) m = min(m, (uint32_t)dpp_i<0x4E>((int)m));
    if constexpr (LL >= 3) m = min(m, (uint32_t)dpp_i<0x141>((int)m));
    if constexpr (LL >= 4) m = min(m, (uint32_t)dpp_i<0x140>((int)m));
    if constexpr (LL >= 5) m = min(m, (uint32_t)__shfl_xor((int)m, 16));
    if constexpr (LL >= 6) m = min(m, (uint32_t)__shfl_xor((int)m, 32));
    return m;
}

template <int V>
struct VecLoad;
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
};
template <>
struct VecLoad<2> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[2])
    {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4])
    {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// The kernels carry a `dbg` word whose bits switch parts of them off for timing experiments
// (tools/exp_*.sh).  Only a build with ODBOX_TUNING_HOOKS honours it; in the product build the word is
// forced to 0 at the top of each kernel, so those branches are dead code and cannot be reached.
#ifdef ODBOX_TUNING_HOOKS
#define ODBOX_NO_EXPERIMENTS(word) ((void)0)
#else
#define ODBOX_NO_EXPERIMENTS(word) ((word) = 0u)
#endif

// Phase stamps of the tail kernels, diagnostic builds only (-DODBOX_STAMPS, tools/exp_nms_stamps.py): every wave (or
// block) adds the length of each of its phases, in 10 ns ticks of the constant-rate clock, to a sum and a maximum.
// Each stamp drains the wave's memory operations first, so such a build is slower and only the SHARES mean anything.
#ifdef ODBOX_STAMPS
__device__ unsigned long long g_stamp_sum[64], g_stamp_max[64], g_stamp_cnt[64];
struct Stamps {
    unsigned long long t;
    __device__ __forceinline__ Stamps() : t(now()) {}
    static __device__ __forceinline__ unsigned long long now()
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        return wall_clock64();
    }
    __device__ __forceinline__ void lap(int slot, bool leader)
    {
        const unsigned long long n = now();
        if (leader) {
            atomicAdd(&g_stamp_sum[slot], n - t);
            atomicMax(&g_stamp_max[slot], n - t);
            atomicAdd(&g_stamp_cnt[slot], 1ull);
        }
        t = now();
    }
};
#define ODBOX_LAP(st, slot, leader) (st).lap((slot), (leader))
#else
struct Stamps {};
#define ODBOX_LAP(st, slot, leader) ((void)0)
#endif

constexpr uint32_t SCAN_QUEUES = 64;       // ticket counters (tile queues)
constexpr uint32_t SCAN_TICKET_STRIDE = 64;   // ints between counters: atomics on one cache line serialise
constexpr uint32_t SCAN_MAX_ROWS = 512;    // rows per tile; also the size of the pending-key list

// 16-bit scores (HALF; odbox_decode_nms_mixed): the [B*N, C] matrix of binary16 / bfloat16 scores is read as a
// [B*N, C/2] matrix of dwords, two classes per dword (C even), so tiles, DMA and LDS reads are the fp32 kernel's
// with half the columns.  Both formats are sign-magnitude with the exponent above the mantissa: as SIGNED 16-bit
// integers every positive value is above every negative one and positive values keep their order, so with a
// threshold >= 0 the row maximum that matters (a row counts only if its maximum exceeds the threshold, hence is
// positive) is a packed integer maximum of the raw bits (v_pk_max_i16: one instruction per two classes, the
// fp32 kernel's rate per byte).  A negative threshold takes the general form: x ^ ((x >> 15) & 0x7fff) maps every
// finite value to a two's complement integer of the same order (the map is its own inverse).  The maximum is
// widened exactly (fp16: v_cvt_f32_f16, bf16: a shift) and from there on the kernel is the fp32 one: same keys.
typedef short v2i16 __attribute__((ext_vector_type(2)));
enum : uint32_t { HALF_BF16 = 1u, HALF_GENERAL = 2u };
__device__ __forceinline__ uint32_t h16_ordered(uint32_t x)          // two halves at once, involutive
{
    const v2i16 sh = {15, 15};
    const uint32_t sign = __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2i16, x) >> sh);    // 0xffff where negative
    return x ^ (sign & 0x7fff7fffu);
}
__device__ __forceinline__ uint32_t h16_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2i16, a), __builtin_bit_cast(v2i16, b)));
}
__device__ __forceinline__ float h16_widen(uint32_t bits16, bool bf16)
{
    return bf16 ? __uint_as_float(bits16 << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)bits16);
}

// The call's counters are zeroed INSIDE its first kernel (round 3; a zero_header_kernel in front of the scan cost
// 4.6 us plus a kernel boundary on every call).  The workspace starts with two 64-bit words on a 256-byte line of
// their own: `epoch`, which the last kernel of every call raises by one, and `ready`.  Block 0 of the scan zeroes
// every counter of the call (write-through stores, drained) and then stores ready = epoch + 1; every other wave
// polls `ready` for that value before its first atomic on a counter -- its first tile is assigned statically and
// its DMA is already in flight by then.  Both words live in device memory, so a replayed HIP graph works like an
// eager call (a per-call host value would be frozen at capture).  A workspace with arbitrary contents is fine:
// the stale `ready` of an earlier call is epoch, not epoch + 1, and garbage equals epoch + 1 with probability
// 2^-64.  Only a call whose launches fail half-way (ODBOX_EHIP) leaves the two words in an undefined relation:
// zero the first 16 bytes of the workspace, or take a fresh one, before using it again (include/odbox.h).
constexpr size_t HDR_BYTES = 256;
typedef unsigned long long hdr_word;
__device__ __forceinline__ void wait_ready(hdr_word *hdr, hdr_word want)
{
    // bounded: ~0.2 s of polling, then go on (wrong detections, never a hang)
    for (uint32_t spins = 0; spins < (1u << 17); ++spins) {
        if (__hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

template <int V, int LL, bool HALF>
__global__ __launch_bounds__(WAVE) void score_scan_kernel(hdr_word *hdr, uint32_t B, int *coop_count,
    const float *__restrict__ scores, int N, int C, ScanGeom g, float score_thr, uint32_t tiles_per_image,
    uint32_t total_tiles, int *__restrict__ tile_ticket, int *__restrict__ cand_count,
    uint64_t *__restrict__ cand_keys, uint32_t cand_stride, uint32_t nq_arg, uint32_t half_mode, uint32_t dbg)
{
    // (HALF: `scores` points at 16-bit data and C counts DWORDS per row, i.e. classes / 2)
    ODBOX_NO_EXPERIMENTS(dbg);
    extern __shared__ __attribute__((aligned(16))) float s_buf[];     // 2 tile buffers, then the pending keys
    constexpr uint32_t LPR = 1u << LL;                  // lanes per row
    constexpr uint32_t G = WAVE >> LL;                  // rows per wave-step
    const uint32_t lane = threadIdx.x;
    const uint32_t sub = lane & (LPR - 1);
    const uint32_t grp = lane >> LL;
    const uint32_t R = (uint32_t)g.steps * G;
    uint64_t *s_pend = reinterpret_cast<uint64_t *>(s_buf + 2u * g.buf_floats);   // [R] keys of the tile just reduced
    const uint32_t lds_base = lds_address(s_buf);
    const uint32_t stride_k = LPR * V;                  // floats between a lane's consecutive reads

    auto tile_origin = [&](uint32_t t, uint32_t &b, uint32_t &n0) {
        divmod_u32(t, tiles_per_image, g.tpi_magic, b, n0);
        n0 *= R;
    };
    auto issue = [&](uint32_t t, uint32_t which) {
        uint32_t b, n0;
        tile_origin(t, b, n0);
        const uint32_t rows = min(R, (uint32_t)N - n0);
        const float *src = scores + ((size_t)b * N + n0) * (size_t)C;
        const uint32_t dst = lds_base + which * g.buf_floats * 4u;
        if (((uintptr_t)src & 15u) == 0)
            dma_tile<4>(src, dst, rows * (uint32_t)C, lane, g.stream != 0);
        else
            dma_tile<1>(src, dst, rows * (uint32_t)C, lane, g.stream != 0);
    };

    // Tiles are handed out by ticket counters: a workgroup that is not resident at launch finds
    // its queue empty instead of stretching the tail.  One counter saturates near 88 tickets/us
    // (MI355X_MICROARCH.md "dequeue"), far below the tile rate of this kernel, so the tile
    // range is cut into SCAN_QUEUES contiguous queues with one counter (on its own cache line)
    // each; block i draws from queue i % Q and its first tile is its rank i / Q in that queue.
    // vmcnt retires in order, so anything that waits on a vector-memory result issued AFTER the
    // prefetch DMAs would wait for the whole prefetch.  Hence: both returning atomics of an
    // iteration (the ticket for the tile after next, the slot reservation for the candidates
    // of the previous tile) are issued at the top, BEFORE the iteration's one s_waitcnt
    // vmcnt(0), where only the current tile is outstanding; the DMAs of the next tile go out
    // after it; and the reduce phase issues no vector-memory instruction at all.
    const hdr_word want = hdr[0] + 1ull;                              // written by the last kernel of the previous call
    if (blockIdx.x == 0) {
        // cand_count[B] and fin_count[B] (one 256-byte line each, adjacent arrays), the tile tickets, coop_count
        for (uint32_t i = lane; i < 2u * B; i += WAVE)
            __hip_atomic_store(cand_count + (size_t)i * CNT_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < SCAN_QUEUES)
            __hip_atomic_store(tile_ticket + lane * SCAN_TICKET_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 3) __hip_atomic_store(coop_count + lane, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one per pass, then the number of images left to pass 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every zero has left this wave
        if (lane == 0) __hip_atomic_store(hdr + 1, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t nq = min(nq_arg, gridDim.x);                       // every queue has a block
    const uint32_t q = blockIdx.x % nq;
    const uint32_t q_lo = (uint32_t)(((uint64_t)total_tiles * q) / nq);
    const uint32_t q_hi = (uint32_t)(((uint64_t)total_tiles * (q + 1)) / nq);
    const uint32_t q_blocks = (gridDim.x - q + nq - 1) / nq;          // blocks drawing from queue q
    int *my_ticket = tile_ticket + q * SCAN_TICKET_STRIDE;
    auto draw = [&]() -> uint32_t {
        int v = 0;
        if (lane == 0) v = atomicAdd(my_ticket, 1);
        return q_lo + q_blocks + (uint32_t)__builtin_amdgcn_readfirstlane(v);
    };
    uint32_t t = q_lo + blockIdx.x / nq;
    if (t >= q_hi) return;
    issue(t, 0);
    if (blockIdx.x != 0) wait_ready(hdr, want);                       // the counters are zero from here on
    uint32_t nt = draw();
    uint32_t pend_n = 0, pend_b = 0;

    for (uint32_t it = 0;; ++it) {
        const uint32_t nnt = nt < q_hi ? draw() : nt;                 // ticket for the tile after next
        uint32_t slot = 0;
        if (pend_n != 0) {
            int v = 0;
            if (lane == 0) v = atomicAdd(cand_count + pend_b * CNT_STRIDE, (int)pend_n);
            slot = (uint32_t)__builtin_amdgcn_readfirstlane(v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // tile t has landed
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < pend_n; i += WAVE)
            if (slot + i < cand_stride) cand_keys[(size_t)pend_b * cand_stride + slot + i] = s_pend[i];
        if (nt < q_hi && !(dbg & 2u)) issue(nt, (it + 1) & 1u);

        // ---- reduce tile t out of LDS (no vector-memory instruction in here) ----------------
        const float *buf = s_buf + (it & 1u) * g.buf_floats;
        uint32_t b, n0;
        tile_origin(t, b, n0);
        const uint32_t rows = min(R, (uint32_t)N - n0);
        const uint32_t nsteps = (dbg & 1u) ? 0u : (rows + G - 1) >> (6 - LL);
        pend_n = 0;
        pend_b = b;
        for (uint32_t step = 0; step < nsteps; ++step) {
            const uint32_t row = step * G + grp;
            const bool live = row < rows;
            // lanes past the last row read the last row (valid address) and drop the result
            const float *rowp = buf + min(row, rows - 1) * (uint32_t)C + sub * V;
            float m;
            if constexpr (!HALF) {
                float m0 = -INFINITY, m1 = -INFINITY;
                int k = 0;
#pragma unroll 4
                for (; k + 2 <= g.kfull; k += 2) {
                    float v0[V], v1[V];
                    VecLoad<V>::load(rowp + (uint32_t)k * stride_k, v0);
                    VecLoad<V>::load(rowp + (uint32_t)(k + 1) * stride_k, v1);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        m0 = fmaxf(m0, v0[e]);
                        m1 = fmaxf(m1, v1[e]);
                    }
                }
                for (; k < g.K; ++k) {                       // odd read and the partial last read
                    const bool ok = ((uint32_t)k * LPR + sub) * V < (uint32_t)C;
                    float v0[V];
                    VecLoad<V>::load(ok ? rowp + (uint32_t)k * stride_k : rowp, v0);
#pragma unroll
                    for (int e = 0; e < V; ++e) m0 = fmaxf(m0, ok ? v0[e] : -INFINITY);
                }
                m = group_max<LL>(fmaxf(m0, m1));
            } else {
                // packed signed-integer maxima of the raw (or order-mapped) bits; 0x8000 = the smallest int16
                auto fold_row = [&](auto general_c) {
                    constexpr bool GENERAL = decltype(general_c)::value;
                    uint32_t a0 = 0x80008000u, a1 = 0x80008000u;
                    int k = 0;
#pragma unroll 8
                    for (; k + 2 <= g.kfull; k += 2) {
                        float v0[V], v1[V];
                        VecLoad<V>::load(rowp + (uint32_t)k * stride_k, v0);
                        VecLoad<V>::load(rowp + (uint32_t)(k + 1) * stride_k, v1);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const uint32_t x0 = __float_as_uint(v0[e]), x1 = __float_as_uint(v1[e]);
                            a0 = h16_max(a0, GENERAL ? h16_ordered(x0) : x0);
                            a1 = h16_max(a1, GENERAL ? h16_ordered(x1) : x1);
                        }
                    }
                    for (; k < g.K; ++k) {
                        const bool ok = ((uint32_t)k * LPR + sub) * V < (uint32_t)C;
                        float v0[V];
                        VecLoad<V>::load(ok ? rowp + (uint32_t)k * stride_k : rowp, v0);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const uint32_t x0 = __float_as_uint(v0[e]);
                            a0 = h16_max(a0, ok ? (GENERAL ? h16_ordered(x0) : x0) : 0x80008000u);
                        }
                    }
                    a0 = h16_max(a0, a1);
                    a0 = h16_max(a0, a0 >> 16) & 0xffffu;                      // the larger of the two halves
                    // general form: no finite value maps to 0x8000, so that is "this lane read nothing": -inf
                    if (GENERAL) a0 = a0 == 0x8000u ? ((half_mode & HALF_BF16) ? 0xff80u : 0xfc00u) : (h16_ordered(a0) & 0xffffu);
                    return a0;
                };
                const uint32_t top = (half_mode & HALF_GENERAL) ? fold_row(std::true_type{}) : fold_row(std::false_type{});
                m = group_max<LL>(h16_widen(top, (half_mode & HALF_BF16) != 0u));
            }
            const bool cand = live && (m > score_thr);                          // bndbox.py:205
            if (__ballot(cand) == 0ull) continue;                               // wave-uniform, common
            // rare: find tf.argmax's FIRST maximal column of the candidate rows, queue the keys
            // (a wave-cooperative walk of each candidate row -- 64 columns per read, a ballot for the first hit --
            // was measured in round 2: 176 -> 197 us at config 5, the ballots serialise; per-lane rows stay)
            uint32_t bcol = 0xffffffffu;
            if (cand) {
                auto check = [&](uint32_t kk, const float (&v0)[V]) {
                    const uint32_t col = (kk * LPR + sub) * V;
#pragma unroll
                    for (int e = V - 1; e >= 0; --e) {
                        if constexpr (!HALF) {
                            if (v0[e] == m) bcol = min(bcol, col + (uint32_t)e);
                        } else {                         // dword col + e holds classes 2 (col + e) and 2 (col + e) + 1
                            const uint32_t x = __float_as_uint(v0[e]);
                            const bool bf = (half_mode & HALF_BF16) != 0u;
                            if (h16_widen(x >> 16, bf) == m) bcol = min(bcol, 2u * (col + (uint32_t)e) + 1u);
                            if (h16_widen(x & 0xffffu, bf) == m) bcol = min(bcol, 2u * (col + (uint32_t)e));
                        }
                    }
                };
                // SCAN_UB reads issued before the first compare: one LDS round trip per batch, not per read (at 4 500
                // candidates per image nearly every step has a candidate row, and with 16-bit scores -- 45 dword
                // reads per row at C = 90 -- the read-wait-compare form of this loop was what bounded the kernel)
                constexpr int SCAN_UB = V == 4 ? 8 : 16;
                int kk = 0;
                for (; kk + SCAN_UB <= g.kfull; kk += SCAN_UB) {
                    float vv[SCAN_UB][V];
#pragma unroll
                    for (int j = 0; j < SCAN_UB; ++j) VecLoad<V>::load(rowp + (uint32_t)(kk + j) * stride_k, vv[j]);
#pragma unroll
                    for (int j = 0; j < SCAN_UB; ++j) check((uint32_t)(kk + j), vv[j]);
                }
                for (; kk < g.K; ++kk) {
                    if (((uint32_t)kk * LPR + sub) * V < (uint32_t)C) {
                        float v0[V];
                        VecLoad<V>::load(rowp + (uint32_t)kk * stride_k, v0);
                        check((uint32_t)kk, v0);
                    }
                }
            }
            bcol = group_min<LL>(bcol);
            const uint64_t mask = __ballot(cand && sub == 0);
            if (cand && sub == 0)
                s_pend[pend_n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = cand_key(bcol, m, n0 + row);
            pend_n += (uint32_t)__popcll(mask);
        }
        if (nt >= q_hi) break;
        t = nt;
        nt = nnt;
    }
    if (pend_n != 0) {                                  // candidates of the last tile
        int slot_raw = 0;
        if (lane == 0) slot_raw = atomicAdd(cand_count + pend_b * CNT_STRIDE, (int)pend_n);
        const uint32_t base = (uint32_t)__shfl(slot_raw, 0);
        for (uint32_t i = lane; i < pend_n; i += WAVE)
            if (base + i < cand_stride) cand_keys[(size_t)pend_b * cand_stride + base + i] = s_pend[i];
    }
}

// ------------------------------------------------------------------------------------------
// In-place ascending sort of n u64 keys (any n) with the "flip" bitonic network: every
// compare-exchange moves the minimum to the lower index, so the virtual +inf padding above n
// never moves and needs no storage.  BLOCK: all threads of the block, __syncthreads between
// stages; otherwise the 64 lanes of one wave on wave-private memory.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <bool BLOCK>
__device__ __forceinline__ void sort_sync()
{
    if constexpr (BLOCK) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

template <bool BLOCK, typename KeyPtr>
__device__ __forceinline__ void sort_u64(KeyPtr keys, uint32_t n, uint32_t tid, uint32_t nthreads)
{
    if (n < 2) return;
    const uint32_t half = next_pow2(n) >> 1;
    for (uint32_t k = 2; (k >> 1) <= half; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = tid; t < half; t += nthreads) {          // flip stage
            const uint32_t base = (t & ~(hk - 1)) << 1, off = t & (hk - 1);
            const uint32_t lo = base + off, hi = base + k - 1 - off;
            if (hi < n) {
                const uint64_t x = keys[lo], y = keys[hi];
                if (x > y) { keys[lo] = y; keys[hi] = x; }
            }
        }
        sort_sync<BLOCK>();
        for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < half; t += nthreads) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                if (hi < n) {
                    const uint64_t x = keys[lo], y = keys[hi];
                    if (x > y) { keys[lo] = y; keys[hi] = x; }
                }
            }
            sort_sync<BLOCK>();
        }
    }
}

// 64 keys, one per lane, sorted across the wave in registers (same network, shuffles)
__device__ __forceinline__ uint64_t wave_sort64(uint64_t key, uint32_t lane)
{
    for (uint32_t k = 2; k <= WAVE; k <<= 1) {
        {
            const uint64_t other = __shfl(key, (int)(lane ^ (k - 1)));
            const bool lower = (lane & (k >> 1)) == 0;
            key = lower ? (other < key ? other : key) : (other > key ? other : key);
        }
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            const uint64_t other = __shfl(key, (int)(lane ^ j));
            const bool lower = (lane & j) == 0;
            key = lower ? (other < key ? other : key) : (other > key ? other : key);
        }
    }
    return key;
}

// 64 * KPL keys sorted across the wave in registers, element e = lane * KPL + slot (the same
// flip network; stages whose distance is below KPL stay inside a lane, the others exchange
// registers with the partner lane).  MERGE_ONLY: the input is one bitonic sequence and only
// the last merge (half-cleaners 32*KPL .. 1) runs.
template <int KPL, bool MERGE_ONLY = false>
__device__ __forceinline__ void wave_sort_regs(uint64_t (&k)[KPL], uint32_t lane)
{
    constexpr uint32_t P = WAVE * KPL;
    auto cmpswap = [](uint64_t &lo, uint64_t &hi) {
        const uint64_t x = lo, y = hi;
        lo = x < y ? x : y;
        hi = x < y ? y : x;
    };
    auto keep = [](uint64_t mine, uint64_t other, bool lower) {
        const bool take_other = lower ? (other < mine) : (other > mine);
        return take_other ? other : mine;
    };
    constexpr int LOGP = 6 + (KPL == 1 ? 0 : KPL == 2 ? 1 : KPL == 4 ? 2 : 3);
#pragma unroll
    for (int lk = MERGE_ONLY ? LOGP : 1; lk <= LOGP; ++lk) {
        const uint32_t kk = 1u << lk;
        if (!MERGE_ONLY) {                       // flip stage: partner e ^ (kk - 1)
            if (kk <= (uint32_t)KPL) {
#pragma unroll
                for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0)
                    if ((s0 ^ (kk - 1)) > s0) cmpswap(k[s0], k[s0 ^ (kk - 1)]);
            } else {
                const uint32_t lp = lane ^ (kk / KPL - 1);
                const bool lower = (lane & (kk / (2 * KPL))) == 0;
                uint64_t other[KPL];
#pragma unroll
                for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0) other[s0] = __shfl(k[KPL - 1 - s0], (int)lp);
#pragma unroll
                for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0) k[s0] = keep(k[s0], other[s0], lower);
            }
        }
#pragma unroll
        for (int lj = MERGE_ONLY ? lk - 1 : lk - 2; lj >= 0; --lj) {           // half-cleaners: partner e ^ j
            const uint32_t j = 1u << lj;
            if (j < (uint32_t)KPL) {
#pragma unroll
                for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0)
                    if ((s0 & j) == 0) cmpswap(k[s0], k[s0 | j]);
            } else {
                const uint32_t lp = lane ^ (j / KPL);
                const bool lower = (lane & (j / KPL)) == 0;
#pragma unroll
                for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0) k[s0] = keep(k[s0], __shfl(k[s0], (int)lp), lower);
            }
        }
    }
}

// sort n <= 64 * KPL keys of `src` (any memory) into the wave's LDS array `dst`
template <int KPL>
__device__ __forceinline__ void wave_sort_to_lds(const uint64_t *src, uint32_t n, uint64_t *dst, uint32_t lane)
{
    uint64_t k[KPL];
#pragma unroll
    for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0) {
        const uint32_t e = lane * KPL + s0;
        k[s0] = e < n ? src[e] : ~0ull;
    }
    wave_sort_regs<KPL>(k, lane);
#pragma unroll
    for (uint32_t s0 = 0; s0 < (uint32_t)KPL; ++s0) {
        const uint32_t e = lane * KPL + s0;
        if (e < n) dst[e] = k[s0];
    }
}

// The three kernels behind the scan (partition, NMS, finalize) keep their LDS small enough to fit on a CU beside the
// resident scan waves of another stream's batch.  The scan leaves TAIL_LDS_RESERVE bytes by its own count, but LDS is
// handed out in granules coarser than 1 KB (1 280 B fits every observation): six scan waves of 23 552 B take
// 6 x 24 320 B, and what is really left of the 160 KB is 17 920 B.  A 16 KB partition block fitted (round 2, 182 us
// per batch with two batches in flight), an 18.8 KB one -- histogram and sort buffer side by side -- did not, and
// every tail kernel waited for the other stream's scan to END (207 us; same-box bisection of two commits).  Hence
// PART_SORT_CAP: histogram (2.4 KB at C = 90) + 12 KB of sort buffer = 14.7 KB.
#ifndef ODBOX_TAIL_LDS_KB
#define ODBOX_TAIL_LDS_KB 22
#endif
constexpr size_t TAIL_LDS_RESERVE = ODBOX_TAIL_LDS_KB * 1024;
constexpr uint32_t TAIL_SORT_CAP = 2048;   // keys the finalize blocks sort in LDS (16 KiB); more: in global memory
#ifndef ODBOX_PART_SORT_CAP
#define ODBOX_PART_SORT_CAP 1536
#endif
constexpr uint32_t PART_SORT_CAP = ODBOX_PART_SORT_CAP;   // keys the partition blocks sort in LDS (12 KiB), beside their histogram
constexpr uint32_t NMS_SEG_LDS = 512;      // keys one NMS wave sorts by itself; bigger segments: block sort
#ifndef ODBOX_NMS_COOP_MIN
#define ODBOX_NMS_COOP_MIN 128
#endif
constexpr int NMS_COOP_MIN = ODBOX_NMS_COOP_MIN;          // segments above this (and <= NMS_SEG_LDS): nms_coop_kernel, one block each

// kernel 2: counting sort of one image's candidate keys by class (LDS histogram -> scan ->
// scatter); writes the class segment starts.  A class with more keys than an NMS block sorts on its
// own (NMS_SEG_LDS) is put in visiting order here, segment by segment (in LDS up to PART_SORT_CAP keys,
// in global memory beyond): the NMS kernel takes every segment above NMS_SEG_LDS as presorted.
#ifndef ODBOX_PART_THREADS
#define ODBOX_PART_THREADS 512
#endif
constexpr int PART_THREADS = ODBOX_PART_THREADS;

// The class histogram is kept in PART_COPIES copies, one per lane & (PART_COPIES - 1): candidates arrive in runs of one
// class (neighbouring anchors of one object), and lanes that add to one LDS word are served one after the other --
// 64 per instruction with a single copy, 8 with eight (stamps, one image: histogram 3.0 -> 1.6 us, scatter 2.3 ->
// 1.3 us).  Many classes: one copy (the copies must fit beside the sort buffer).
constexpr int PART_COPIES = 8;
__host__ __device__ constexpr int part_copies(int C) { return C <= 128 ? PART_COPIES : 1; }
__host__ __device__ constexpr size_t part_sort_offset(int C)      // byte offset of the sort buffer behind hist + scan scratch
{
    return ((size_t)(C * part_copies(C) + PART_THREADS + 1) * sizeof(int) + 15) & ~(size_t)15;
}
// what six resident scan waves really leave of a CU's LDS (see TAIL_LDS_RESERVE): the tail kernels must fit into it at
// the configuration the pipeline is measured on, or two batches in flight silently stop overlapping
constexpr size_t TAIL_LDS_FIT = 17920;
static_assert(part_sort_offset(90) + (size_t)PART_SORT_CAP * 8 <= TAIL_LDS_FIT, "partition block does not fit beside a scan");
static_assert((size_t)TAIL_SORT_CAP * 8 <= TAIL_LDS_FIT, "finalize block does not fit beside a scan");

//
// Pre-NMS top-k (an extension, SURVEY.md 8d: the reference has no such bound; pre_nms_topk = 0 turns it off):
// keep the pre_nms_topk highest scores of the image, ties to the lower anchor, BEFORE the class partition.
// An image whose list is not longer than the bound -- BASELINE configs[4]: 4 500 candidates against 5 000 --
// pays one compare for it.  A longer list is first sorted by (score desc, anchor asc), in the block's LDS
// sort buffer up to PART_SORT_CAP keys and in global memory beyond, and cut to the bound; the survivors
// then go through the partition below like any other list.
//
// The score tier (round 3).  Greedy NMS visits a class in score order and a candidate is only ever suppressed by
// higher-scoring ones, so the kept boxes with score >= t are decided by the candidates with score >= t alone -- and
// the final list is the max_ret best kept boxes of the image.  Hence: pass 1 runs partition / NMS / final top-k on the
// image's TIER only, every candidate with score >= t where t is chosen so that the tier holds about TIER_FACTOR *
// max_ret of them (all ties at t included); if the tier keeps at least max_ret boxes, those ARE the image's
// detections, bit for bit, and the rest of its candidates (4 500 per image on configs[4], of which the tier is 800)
// never reach the NMS kernel: its segments shrink from up to 330 candidates to at most a wave's worth, and the
// latency chain that bounds the launch (the longest class) with them.  An image whose tier keeps fewer than max_ret
// boxes -- or whose list the pre-NMS top-k cuts, or is longer than the block holds in registers -- is done again by
// pass 2 on its whole list, exactly as before; the three pass-2 launches are always enqueued (no host round trip)
// and their blocks leave at once for images that are done (state[b] != 0).
#ifndef ODBOX_TIER_FACTOR
#define ODBOX_TIER_FACTOR 8
#endif
} // namespace
static std::atomic<int> g_tier_factor{ODBOX_TIER_FACTOR};      // odbox_set_nms_tier
static std::atomic<int> g_fused_tail{1};                       // odbox_set_nms_fused_tail
namespace {
constexpr uint32_t TIER_BUCKETS = 1024;                   // score histogram of the tier selection (4 KB of LDS)

constexpr int PART_REG = 16;                              // keys per thread that a partition block holds in registers

// Wavefront reductions and scans by DPP: no LDS round trips (a __shfl is a ds_bpermute, and six dependent ones are a
// third of a microsecond in blocks whose whole life is ten).  Lanes a control leaves unwritten take `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = false>
__device__ __forceinline__ int dpp_i(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, BANK_MASK, BOUND); }
// inclusive prefix sum over the 64 lanes: three shifts inside the quads' reach, rows of 16, then row 0 -> 1, 2 -> 3, 1 -> 2 and 3
__device__ __forceinline__ int wave_incl_scan(int x)
{
    int s = x + dpp_i<0x111, 0xf, 0xf, true>(0, x);        // row_shr:1
    s += dpp_i<0x112, 0xf, 0xf, true>(0, x);               // row_shr:2
    s += dpp_i<0x113, 0xf, 0xf, true>(0, x);               // row_shr:3
    s += dpp_i<0x114, 0xf, 0xe>(0, s);                     // row_shr:4, lanes 4..15 of a row
    s += dpp_i<0x118, 0xf, 0xc>(0, s);                     // row_shr:8, lanes 8..15
    s += dpp_i<0x142, 0xa, 0xf>(0, s);                     // row_bcast:15 into rows 1 and 3
    s += dpp_i<0x143, 0xc, 0xf>(0, s);                     // row_bcast:31 into rows 2 and 3
    return s;
}
// minimum / maximum over the 64 lanes, wave-uniform: butterflies inside a row of 16, then the four rows through SGPRs
template <bool MAX>
__device__ __forceinline__ uint32_t wave_minmax_u32(uint32_t x)
{
    auto op = [](uint32_t a, uint32_t b) { return MAX ? max(a, b) : min(a, b); };
    x = op(x, (uint32_t)dpp_i<0xB1, 0xf, 0xf, true>(0, (int)x));      // quad_perm [1,0,3,2]
    x = op(x, (uint32_t)dpp_i<0x4E, 0xf, 0xf, true>(0, (int)x));      // quad_perm [2,3,0,1]
    x = op(x, (uint32_t)dpp_i<0x141, 0xf, 0xf, true>(0, (int)x));     // row_half_mirror
    x = op(x, (uint32_t)dpp_i<0x140, 0xf, 0xf, true>(0, (int)x));     // row_mirror
    return op(op((uint32_t)__builtin_amdgcn_readlane((int)x, 0), (uint32_t)__builtin_amdgcn_readlane((int)x, 16)),
              op((uint32_t)__builtin_amdgcn_readlane((int)x, 32), (uint32_t)__builtin_amdgcn_readlane((int)x, 48)));
}
// sum of s[0 .. n) for a wave-uniform n <= NW: NW independent broadcast reads issued together instead of n dependent trips
template <int NW>
__device__ __forceinline__ int sum_first(const int *s, uint32_t n)
{
    int v[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) v[q] = s[q];
    int acc = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) acc += (uint32_t)q < n ? v[q] : 0;
    return acc;
}

// The tier of one image, by the whole block (THREADS threads; kreg = the image's keys, ~0 = none): marks every key
// outside the tier ~0 and returns the tier's size.  The cut is found without a sort: a TIER_BUCKETS-bucket histogram
// over the top bits of the (order-preserving) score word, the bucket in which the running count passes tier_k taken
// whole (ties included, so the tier is a score threshold, not a count).  s_tier: TIER_BUCKETS + 2 ints of LDS, s_scan:
// 2 * THREADS / WAVE ints; the caller puts a barrier behind the call before it reuses either.
template <int THREADS, int REG>
__device__ __forceinline__ uint32_t tier_select(uint64_t (&kreg)[REG], uint32_t M, uint32_t tier_k, int *s_tier, int *s_scan,
                                                uint32_t tid)
{
    constexpr int NW = THREADS / WAVE;                   // waves of the block
    constexpr uint32_t BPT = TIER_BUCKETS / THREADS;     // score buckets per thread in the scan
    static_assert(BPT * THREADS == TIER_BUCKETS && (BPT == 1 || BPT == 2), "one or two score buckets per thread");
    uint32_t lo = 0xffffffffu, hi = 0u;      // the key's score word: smaller = higher score
#pragma unroll
    for (int r = 0; r < REG; ++r)
        if (kreg[r] != ~0ull) {
            const uint32_t sw = (uint32_t)(kreg[r] >> 20);
            lo = min(lo, sw);
            hi = max(hi, sw);
        }
    lo = wave_minmax_u32<false>(lo);
    hi = wave_minmax_u32<true>(hi);
    if ((tid & 63u) == 0u) {
        s_scan[tid >> 6] = (int)lo;
        s_scan[NW + (tid >> 6)] = (int)hi;
    }
    for (uint32_t i = tid; i < TIER_BUCKETS + 2u; i += THREADS) s_tier[i] = 0;
    __syncthreads();
    for (int w = 0; w < NW; ++w) {
        lo = min(lo, (uint32_t)s_scan[w]);
        hi = max(hi, (uint32_t)s_scan[NW + w]);
    }
    const uint32_t range = hi - lo;
    const uint32_t shift = range < TIER_BUCKETS ? 0u : (32u - (uint32_t)__clz((int)range)) - 10u;      // (range >> shift) < 1024
#pragma unroll
    for (int r = 0; r < REG; ++r)
        if (kreg[r] != ~0ull) atomicAdd(&s_tier[((uint32_t)(kreg[r] >> 20) - lo) >> shift], 1);
    __syncthreads();
    // inclusive scan over the buckets, BPT per thread
    const int h0 = s_tier[BPT * tid], h1 = BPT == 2 ? s_tier[BPT * tid + 1u] : 0;
    const int incl = wave_incl_scan(h0 + h1);
    __syncthreads();                         // the lo / hi slots of s_scan have been read by everybody
    if ((tid & 63u) == 63u) s_scan[tid >> 6] = incl;
    __syncthreads();
    const int before = incl - (h0 + h1) + sum_first<NW>(s_scan, tid >> 6);
    if ((uint32_t)before < tier_k) {         // exactly one thread finds the bucket the tier_k-th best falls into
        if ((uint32_t)(before + h0) >= tier_k) {
            s_tier[TIER_BUCKETS] = (int)(BPT * tid);
            s_tier[TIER_BUCKETS + 1u] = before + h0;
        } else if (BPT == 2 && (uint32_t)(before + h0 + h1) >= tier_k) {
            s_tier[TIER_BUCKETS] = (int)(BPT * tid + 1u);
            s_tier[TIER_BUCKETS + 1u] = before + h0 + h1;
        }
    }
    __syncthreads();
    const uint32_t last_bucket = (uint32_t)s_tier[TIER_BUCKETS];
    M = min((uint32_t)s_tier[TIER_BUCKETS + 1u], M);      // the tier's size
#pragma unroll
    for (int r = 0; r < REG; ++r)
        if (kreg[r] != ~0ull && (((uint32_t)(kreg[r] >> 20) - lo) >> shift) > last_bucket) kreg[r] = ~0ull;
    return M;
}

// (a device function of the whole block, PART_THREADS threads: pass 2 of an image runs at the end of the pass-1 final
// kernel's block of that image, see finalize_kernel)
__device__ __forceinline__ void partition_image(
    unsigned char *s_part, uint32_t b, uint32_t total_segments,
    const int *__restrict__ cand_count, uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
    uint32_t cand_stride, int C, uint32_t pre_nms_topk, int *__restrict__ seg_start, int *__restrict__ coop_count,
    int *__restrict__ coop_list, int pass, uint32_t tier_k, int *__restrict__ state, int *__restrict__ fin_count)
{
    const int copies = part_copies(C);
    int *s_hist = reinterpret_cast<int *>(s_part);       // [copies][C] counts -> cursors
    int *s_scan = s_hist + copies * C;                   // [PART_THREADS] scan scratch, [PART_THREADS] = flag
    const uint32_t tid = threadIdx.x;
    int *my_hist = s_hist + (int)(tid & (uint32_t)(copies - 1)) * C;      // this lane's copy
    if (pass == 2 && tid == 0) {
        fin_count[b * CNT_STRIDE] = 0;                    // what pass 1 kept of the image is void
        fin_count[b * CNT_STRIDE + 1] = 0;                // segments of the image that pass 2 has finished (nms_kernel)
    }
    Stamps stp;
    uint32_t M = min((uint32_t)cand_count[b * CNT_STRIDE], cand_stride);
    ODBOX_LAP(stp, 0, tid == 0);                          // count loaded
    uint64_t *in = keys_in + (size_t)b * cand_stride;
    uint64_t *out = keys_out + (size_t)b * cand_stride;
    if (pass == 1 && ((pre_nms_topk != 0u && M > pre_nms_topk) || M > (uint32_t)(PART_REG * PART_THREADS))) {
        // left to pass 2 whole: an empty tier (no segment, nothing kept, not complete)
        int *seg = seg_start + (size_t)b * (C + 1);
        for (int c = tid; c <= C; c += PART_THREADS) seg[c] = 0;
        if (tid == 0) state[b] = 0;
        return;
    }
    if (pre_nms_topk != 0u && M > pre_nms_topk) {        // block-uniform (pass 2 only)
        // (the LDS sort buffer exists when the list can hold an oversized class segment, see the launch)
        uint64_t *s_sel = reinterpret_cast<uint64_t *>(s_part + part_sort_offset(C));
        if (M <= PART_SORT_CAP && cand_stride > NMS_SEG_LDS) {
            for (uint32_t i = tid; i < M; i += PART_THREADS) s_sel[i] = cand_to_topk(in[i]);
            __syncthreads();
            sort_u64<true>(s_sel, M, tid, PART_THREADS);
            for (uint32_t i = tid; i < pre_nms_topk; i += PART_THREADS) in[i] = topk_to_cand(s_sel[i]);
        } else {
            for (uint32_t i = tid; i < M; i += PART_THREADS) in[i] = cand_to_topk(in[i]);
            __syncthreads();
            sort_u64<true>(in, M, tid, PART_THREADS);
            for (uint32_t i = tid; i < pre_nms_topk; i += PART_THREADS) in[i] = topk_to_cand(in[i]);
        }
        M = pre_nms_topk;
        __syncthreads();                                 // the cut list is what every thread reads below
    }
    for (int c = tid; c < copies * C; c += PART_THREADS) s_hist[c] = 0;
    if (tid == 0) s_scan[PART_THREADS] = 0;
    __syncthreads();
    // (class / anchor ids are clamped wherever they index memory: a corrupted workspace must give
    // wrong answers, never a fault)
    const uint32_t cmax = (uint32_t)C - 1u;
    // the usual list (<= PART_REG * 512 keys) stays in registers between the two passes: every load
    // is issued before the first use, and the scatter pass reads nothing from memory
    const bool in_regs = M <= (uint32_t)(PART_REG * PART_THREADS);
    uint64_t kreg[PART_REG];
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < PART_REG; ++r) {
            const uint32_t i = tid + (uint32_t)r * PART_THREADS;
            kreg[r] = i < M ? in[i] : ~0ull;
        }
        if (pass == 1) {
            // ---- the tier: every candidate whose score bucket is at most the one in which the tier_k-th best falls ----
            const bool whole = M <= tier_k;              // block-uniform
            if (!whole) {
                M = tier_select<PART_THREADS, PART_REG>(kreg, M, tier_k, reinterpret_cast<int *>(s_part + part_sort_offset(C)), s_scan, tid);
                __syncthreads();                         // s_scan is used again below
            }
            if (tid == 0) state[b] = whole ? 1 : 0;      // 1: the tier is the whole list, whatever it keeps is final
        }
#pragma unroll
        for (int r = 0; r < PART_REG; ++r)
            if (kreg[r] != ~0ull) atomicAdd(&my_hist[min(key_class(kreg[r]), cmax)], 1);
    } else {
        for (uint32_t i = tid; i < M; i += PART_THREADS) atomicAdd(&my_hist[min(key_class(in[i]), cmax)], 1);
    }
    __syncthreads();
    ODBOX_LAP(stp, 1, tid == 0);                          // keys loaded, histogram built
    // exclusive scan over classes: each thread owns a contiguous chunk; wave scans + 8 wave totals
    const int chunk = (C + PART_THREADS - 1) / PART_THREADS;
    const int c0 = min((int)tid * chunk, C), c1 = min(c0 + chunk, C);
    auto class_count = [&](int c) {
        int n = 0;
        for (int k = 0; k < copies; ++k) n += s_hist[k * C + c];
        return n;
    };
    int local = 0, biggest = 0;
    for (int c = c0; c < c1; ++c) {
        const int n = class_count(c);
        local += n;
        biggest = max(biggest, n);
    }
    if (biggest > (int)NMS_SEG_LDS) s_scan[PART_THREADS] = 1;
    const int incl = wave_incl_scan(local);
    if ((tid & 63u) == 63u) s_scan[tid >> 6] = incl;
    __syncthreads();
    int run = incl - local + sum_first<PART_THREADS / WAVE>(s_scan, tid >> 6);
    int *seg = seg_start + (size_t)b * (C + 1);
    for (int c = c0; c < c1; ++c) {
        seg[c] = run;
        int cnt = 0;
        for (int k = 0; k < copies; ++k) {   // each copy becomes the scatter cursor of its lanes
            const int n = s_hist[k * C + c];
            s_hist[k * C + c] = run + cnt;
            cnt += n;
        }
        // segments of NMS_COOP_MIN < m <= NMS_SEG_LDS candidates go to the cooperative blocks of the NMS kernel
        if (cnt > NMS_COOP_MIN && cnt <= (int)NMS_SEG_LDS) {
            const uint32_t at = (uint32_t)atomicAdd(coop_count, 1);
            if (at < total_segments) coop_list[at] = (int)(b * (uint32_t)C + (uint32_t)c);
        }
        run += cnt;
    }
    if (tid == 0) seg[C] = (int)M;
    const bool big = s_scan[PART_THREADS] != 0;
    __syncthreads();
    ODBOX_LAP(stp, 2, tid == 0);                          // class scan, segment starts, cooperative list
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < PART_REG; ++r)
            if (kreg[r] != ~0ull) {
                const uint32_t at = (uint32_t)atomicAdd(&my_hist[min(key_class(kreg[r]), cmax)], 1);
                if (at < M) out[at] = kreg[r];
            }
    } else {
        for (uint32_t i = tid; i < M; i += PART_THREADS) {
            const uint64_t k = in[i];
            const uint32_t at = (uint32_t)atomicAdd(&my_hist[min(key_class(k), cmax)], 1);
            if (at < M) out[at] = k;
        }
    }
    ODBOX_LAP(stp, 3, tid == 0);                          // scatter
    if (!big) return;                                    // block-uniform
    __syncthreads();                                     // scatter visible to the whole block; the last copy's cursor = end of segment c
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(s_part + part_sort_offset(C));
    const int *s_end = s_hist + (copies - 1) * C;
    for (int c = 0; c < C; ++c) {                        // every test below is block-uniform
        const uint32_t st = c > 0 ? min((uint32_t)s_end[c - 1], M) : 0u;
        const uint32_t en = min((uint32_t)s_end[c], M);
        if (en <= st || en - st <= NMS_SEG_LDS) continue;
        const uint32_t m = en - st;
        if (m <= PART_SORT_CAP) {
            for (uint32_t i = tid; i < m; i += PART_THREADS) s_keys[i] = out[st + i];
            __syncthreads();
            sort_u64<true>(s_keys, m, tid, PART_THREADS);
            for (uint32_t i = tid; i < m; i += PART_THREADS) out[st + i] = s_keys[i];
            __syncthreads();
        } else {
            sort_u64<true>(out + st, m, tid, PART_THREADS);
        }
    }
}

__global__ __launch_bounds__(PART_THREADS) void cand_partition_kernel(
    const int *__restrict__ cand_count, uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
    uint32_t cand_stride, int C, uint32_t pre_nms_topk, int *__restrict__ seg_start, int *__restrict__ coop_count,
    int *__restrict__ coop_list, int pass, uint32_t tier_k, int *__restrict__ state, int *__restrict__ fin_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_part[];
    partition_image(s_part, blockIdx.x, gridDim.x * (uint32_t)C, cand_count, keys_in, keys_out, cand_stride, C, pre_nms_topk,
                    seg_start, coop_count, coop_list, pass, tier_k, state, fin_count);
}

// Where a candidate's regression offsets (or its decoded box, anchors == NULL) come from: [B,N,4] fp32, or --
// odbox_decode_nms_mixed -- binary16 / bfloat16 model outputs, widened exactly on load (kind: ODBOX_DTYPE_*).
struct BoxSrc {
    const void *p;
    int kind;
};
__device__ __forceinline__ float4 load_box(const BoxSrc rb, size_t row)
{
    if (rb.kind == ODBOX_DTYPE_F32) return reinterpret_cast<const float4 *>(rb.p)[row];
    const uint2 d = reinterpret_cast<const uint2 *>(rb.p)[row];
    const bool bf = rb.kind == ODBOX_DTYPE_BF16;
    return make_float4(h16_widen(d.x & 0xffffu, bf), h16_widen(d.x >> 16, bf), h16_widen(d.y & 0xffffu, bf), h16_widen(d.y >> 16, bf));
}

// decoded (+ clipped) box of anchor `idx` of image b: bndbox.py:88-99, :105-115
__device__ __forceinline__ float4 candidate_box(const float4 *__restrict__ anchors, const BoxSrc rb, size_t b, int N,
                                                uint32_t idx, int do_clip, float hi_x, float hi_y)
{
    idx = min(idx, (uint32_t)N - 1u);
    float4 v = load_box(rb, b * (size_t)N + idx);
    if (anchors) v = regress_one(anchors[idx], v);
    if (do_clip) v = clip_one(v, hi_x, hi_y);
    return v;
}

// TF NonMaxSuppressionV3 suppression test on min/max-normalised boxes with precomputed areas
// (third-party kernel semantics; call site bndbox.py:170-175):
//     area_p <= 0 or area_q <= 0  ->  IoU = 0;  suppress iff  fl(inter / fl(fl(pa + qa) - inter)) > thr.
// The division is only evaluated where it can matter: with u the rounded union, inter >
// fl(thr_hi * u) already implies fl(inter / u) > thr and inter < fl(thr_lo * u) implies the
// opposite (thr_hi/lo = thr * (1 +- 2^-20), far outside the 2^-23 rounding of either side), so
// only lanes inside that band (practically none) take the exact IEEE division.  The result is
// bit-identical to always dividing.
struct NmsThr {
    float thr, lo, hi;
};

__device__ __forceinline__ bool nms_suppresses(const float4 p, float pa, float qx0, float qy0, float qx1,
                                               float qy1, float qa, const NmsThr t)
{
    const float ih = fmaxf(fminf(p.w, qy1) - fmaxf(p.y, qy0), 0.0f);
    const float iw = fmaxf(fminf(p.z, qx1) - fmaxf(p.x, qx0), 0.0f);
    const float inter = ih * iw;
    const float uni = (pa + qa) - inter;
    const bool valid = (pa > 0.0f) && (qa > 0.0f);
    const bool above = inter > t.hi * uni;
    const bool below = inter < t.lo * uni;
    bool res = valid && above;
    if (__ballot(valid && !above && !below) != 0ull)        // wave-uniform, almost never taken
        res = valid && (above || (!below && (inter / uni > t.thr)));
    return res;
}

// branch-free form for unrolled loops: accumulates "certainly suppresses" into hit and "inside the
// rounding band" into band; the caller resolves band lanes with nms_suppresses_exact
__device__ __forceinline__ void nms_test(const float4 p, float pa, float qx0, float qy0, float qx1, float qy1,
                                         float qa, const NmsThr t, bool &hit, bool &band)
{
    const float ih = fmaxf(fminf(p.w, qy1) - fmaxf(p.y, qy0), 0.0f);
    const float iw = fmaxf(fminf(p.z, qx1) - fmaxf(p.x, qx0), 0.0f);
    const float inter = ih * iw;
    const float uni = (pa + qa) - inter;
    const bool valid = (pa > 0.0f) && (qa > 0.0f);
    const bool above = inter > t.hi * uni;
    const bool below = inter < t.lo * uni;
    hit = hit || (valid && above);
    band = band || (valid && !above && !below);
}
__device__ __forceinline__ bool nms_suppresses_exact(const float4 p, float pa, float qx0, float qy0, float qx1,
                                                     float qy1, float qa, const NmsThr t)
{
    if (pa <= 0.0f || qa <= 0.0f) return false;
    const float ih = fmaxf(fminf(p.w, qy1) - fmaxf(p.y, qy0), 0.0f);
    const float iw = fmaxf(fminf(p.z, qx1) - fmaxf(p.x, qx0), 0.0f);
    const float inter = ih * iw;
    return inter / ((pa + qa) - inter) > t.thr;
}

// value of lane j (wave-uniform j) through an SGPR (v_readlane_b32), no LDS crossbar trip
__device__ __forceinline__ float bcast_lane(float v, int j)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// kernel 3: greedy NMS, one wave per (image, class) segment.  The wave first puts its segment
// in visiting order (score desc, anchor asc): <= 64 keys in registers, <= NMS_SEG_LDS keys in
// its own LDS, longer segments were sorted by the partition kernel.  Candidates are taken 64 at a
// time (one per lane): tested against the kept list, then resolved inside the chunk with
// ballots (the lowest live lane is kept and suppresses the lanes above it).  The first 128
// kept boxes live in REGISTERS, kept box k in lane k & 63 of register set k >> 6, and are
// broadcast with v_readlane (an SGPR operand, no LDS latency in the serial chain); only a kept
// list longer than that (max_ret > 128) continues in LDS.  Stops at max_ret kept (bndbox.py:172).
constexpr int NMS_WAVES = 8;
#ifndef ODBOX_COOP_BLOCKS
#define ODBOX_COOP_BLOCKS 768                // blocks of the NMS launch that take the big segments cooperatively
#endif
constexpr int NMS_REG_KEPT = 128;

// A kept key into the image's final list.  through (pass 2 only): as an agent-scope store, which is written through
// the XCD's L2 -- pass 2's final top-k of an image runs in whichever block finishes its last segment, on any XCD, and
// with the keys already at the coherence point that block's hand-off is "wait for my stores, add to the image's
// counter" instead of a release fence, i.e. an L2 write-back, per block (all images in pass 2: 272 -> 26x us a call).
__device__ __forceinline__ void put_key(uint64_t *slot, uint64_t key, bool through)
{
    if (through)
        __hip_atomic_store(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *slot = key;
}

__device__ __forceinline__ int nms_wave_segment(
    unsigned char *s_raw, uint32_t block, const float4 *__restrict__ anchors, const BoxSrc rb, int N,
    int C, int do_clip, float hi_x, float hi_y, NmsThr thr, int max_ret, const uint64_t *__restrict__ cand_keys,
    uint32_t cand_stride, const int *__restrict__ seg_start,
    int *__restrict__ fin_count, uint64_t *__restrict__ fin_keys, uint32_t fin_stride,
    uint32_t total_segments, int waves_per_block, uint32_t per_wave_bytes, uint32_t sort_cap, bool coop_on, bool kept_alias,
    const int *__restrict__ done, uint32_t dbg)
{
    ODBOX_NO_EXPERIMENTS(dbg);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t seg_id = block * (uint32_t)waves_per_block + wave;
    // returns the image whose segment this wave has finished (an empty segment is finished too), or -1 when the
    // segment is not this wave's to finish: beyond the batch, an image that pass 1 completed, a cooperative block's
    if (seg_id >= total_segments) return -1;
    Stamps stp;
    const uint32_t b = seg_id / (uint32_t)C;
    const uint32_t c = seg_id - b * (uint32_t)C;
    if (done && done[b] != 0) return -1;                 // pass 2: the image's tier was enough
    const int start = seg_start[(size_t)b * (C + 1) + c];
    const int end = seg_start[(size_t)b * (C + 1) + c + 1];
    if (end <= start) return (int)b;
    const uint32_t m = (uint32_t)(end - start);
    const int sl = m <= 16 ? 8 : m <= WAVE ? 16 : 24;     // stamp slots by segment size: tiny / one chunk / two chunks
    ODBOX_LAP(stp, sl + 0, lane == 0);                    // segment bounds loaded

    // per-wave LDS: [sorted segment keys: sort_cap][kept keys: max_ret][kept boxes and areas
    // beyond the register-resident NMS_REG_KEPT: extra * 20 bytes]
    unsigned char *mine = s_raw + (size_t)wave * per_wave_bytes;
    const int extra = max_ret > NMS_REG_KEPT ? max_ret - NMS_REG_KEPT : 0;
    uint64_t *s_sort = reinterpret_cast<uint64_t *>(mine);
    // kept keys: behind the sort buffer, or ON it when kept_alias says the two are never live together (with the
    // cooperative blocks on, a wave that sorts takes the lean path, which keeps its kept set in registers, and the
    // general path below only ever sees presorted segments, which do not use s_sort)
    uint64_t *s_key = kept_alias ? s_sort : s_sort + sort_cap;
    float4 *s_box = reinterpret_cast<float4 *>(s_key + ((max_ret + 1) & ~1)) - NMS_REG_KEPT;   // indexed by kept
    float *s_area = reinterpret_cast<float *>(s_box + NMS_REG_KEPT + extra) - NMS_REG_KEPT;

    const uint64_t *keys = cand_keys + (size_t)b * cand_stride + start;
    // in visiting order already: every segment longer than a wave sorts itself (sorted by the partition kernel)
    const bool presorted = m > NMS_SEG_LDS;
    if (coop_on && !presorted && m > (uint32_t)NMS_COOP_MIN) return -1;    // a cooperative block's segment
    uint64_t reg_key = ~0ull;
    const uint64_t *sorted = keys;
    if (!presorted && !(dbg & 1u)) {
        if (m <= WAVE) {
            reg_key = wave_sort64(lane < m ? keys[lane] : ~0ull, lane);
        } else {                                    // m <= NMS_SEG_LDS by the partition kernel's flag
            if (m <= 2 * WAVE)
                wave_sort_to_lds<2>(keys, m, s_sort, lane);
            else if (m <= 4 * WAVE)
                wave_sort_to_lds<4>(keys, m, s_sort, lane);
            else
                wave_sort_to_lds<8>(keys, m, s_sort, lane);
            sort_sync<false>();
            sorted = s_sort;
        }
    }

    // ---- at most two chunks (every segment this path sorts itself when the cooperative blocks are
    // on): the kept boxes of chunk 0 stay in the registers of their own lanes and are named by a
    // 64-bit mask -- no copy into the kept-list registers, no LDS, ~28 instructions per kept box
    if (!presorted && m <= 2u * WAVE && !(dbg & 3u)) {
        auto fetch = [&](uint32_t i, uint64_t &key, float4 &nb, float &area) {
            key = 0ull;
            nb = make_float4(0.f, 0.f, 0.f, 0.f);
            area = 0.0f;
            if (i < m) {
                key = m <= WAVE ? reg_key : sorted[i];
                const float4 v = candidate_box(anchors, rb, b, N, key_anchor(key), do_clip, hi_x, hi_y);
                nb = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
                area = (nb.w - nb.y) * (nb.z - nb.x);
            }
        };
        uint64_t key0, key1 = 0ull;
        float4 nb0, nb1 = make_float4(0.f, 0.f, 0.f, 0.f);
        float ar0, ar1 = 0.0f;
        ODBOX_LAP(stp, sl + 1, lane == 0);                                // keys loaded and sorted
        fetch(lane, key0, nb0, ar0);
        if (m > WAVE) fetch(WAVE + lane, key1, nb1, ar1);                 // both gathers in flight together
        ODBOX_LAP(stp, sl + 2, lane == 0);                                // boxes gathered
        unsigned long long kmask0 = 0ull, kmask1 = 0ull;
        int kept = 0;
        unsigned long long mask = __ballot(lane < m);
        while (mask != 0ull && kept < max_ret) {
            const int j = __ffsll(mask) - 1;
            kmask0 |= 1ull << j;
            ++kept;
            const bool hit = nms_suppresses(nb0, ar0, bcast_lane(nb0.x, j), bcast_lane(nb0.y, j), bcast_lane(nb0.z, j),
                                            bcast_lane(nb0.w, j), bcast_lane(ar0, j), thr);
            mask &= ~(__ballot(hit) | ((2ull << j) - 1ull));              // lanes <= j are decided
        }
        if (m > WAVE && kept < max_ret) {
            bool hit = false, band = false;
            for (unsigned long long km = kmask0; km != 0ull; km &= km - 1ull) {
                const int j = __ffsll(km) - 1;
                nms_test(nb1, ar1, bcast_lane(nb0.x, j), bcast_lane(nb0.y, j), bcast_lane(nb0.z, j), bcast_lane(nb0.w, j),
                         bcast_lane(ar0, j), thr, hit, band);
            }
            if (__ballot(band && !hit) != 0ull)                           // wave-uniform, practically never
                for (unsigned long long km = kmask0; km != 0ull; km &= km - 1ull) {
                    const int j = __ffsll(km) - 1;
                    if (nms_suppresses_exact(nb1, ar1, bcast_lane(nb0.x, j), bcast_lane(nb0.y, j), bcast_lane(nb0.z, j),
                                             bcast_lane(nb0.w, j), bcast_lane(ar0, j), thr))
                        hit = true;
                }
            mask = __ballot(WAVE + lane < m && !hit);
            while (mask != 0ull && kept < max_ret) {
                const int j = __ffsll(mask) - 1;
                kmask1 |= 1ull << j;
                ++kept;
                const bool h1 = nms_suppresses(nb1, ar1, bcast_lane(nb1.x, j), bcast_lane(nb1.y, j), bcast_lane(nb1.z, j),
                                               bcast_lane(nb1.w, j), bcast_lane(ar1, j), thr);
                mask &= ~(__ballot(h1) | ((2ull << j) - 1ull));
            }
        }
        ODBOX_LAP(stp, sl + 3, lane == 0);                                // greedy walk
        int base = 0;
        if (lane == 0) base = atomicAdd(fin_count + b * CNT_STRIDE, kept);
        base = __shfl(base, 0);
        ODBOX_LAP(stp, sl + 4, lane == 0);                                // slot reservation
        const unsigned long long below = (1ull << lane) - 1ull;
        if ((kmask0 >> lane) & 1ull) {
            const uint32_t pos = (uint32_t)(base + __popcll(kmask0 & below));
            if (pos < fin_stride) put_key(fin_keys + (size_t)b * fin_stride + pos, cand_to_final(key0), done != nullptr);
        }
        if ((kmask1 >> lane) & 1ull) {
            const uint32_t pos = (uint32_t)(base + __popcll(kmask0) + __popcll(kmask1 & below));
            if (pos < fin_stride) put_key(fin_keys + (size_t)b * fin_stride + pos, cand_to_final(key1), done != nullptr);
        }
        return (int)b;
    }

    // kept boxes 0..127: lane k & 63 of set A (k < 64) or set B
    float ax0 = 0.f, ay0 = 0.f, ax1 = 0.f, ay1 = 0.f, aa = 0.f;
    float bx0 = 0.f, by0 = 0.f, bx1 = 0.f, by1 = 0.f, ba = 0.f;
    int kept = 0;
    for (uint32_t base = 0; base < m && kept < max_ret && !(dbg & 2u); base += WAVE) {
        const uint32_t i = base + lane;
        const bool valid = i < m;
        uint64_t key = 0ull;
        if (valid) key = (!presorted && m <= WAVE && !(dbg & 1u)) ? reg_key : sorted[i];
        float4 nb = make_float4(0.f, 0.f, 0.f, 0.f);
        float area = 0.0f;
        if (valid) {
            const float4 v = candidate_box(anchors, rb, b, N, key_anchor(key), do_clip, hi_x, hi_y);
            nb = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
            area = (nb.w - nb.y) * (nb.z - nb.x);
        }
        // (1) against the kept list as it stands at the start of the chunk
        bool alive = valid;
        const int kA = min(kept, WAVE), kB = min(kept, NMS_REG_KEPT);
        // four kept boxes per iteration: independent chains hide the readlane / compare latency
        // of the serial form (one segment of 245 candidates: 30 us -> see profiles/README.md)
        {
            bool hit = false;
            auto group4 = [&](const float rx0, const float ry0, const float rx1, const float ry1, const float ra,
                              int k) {
                bool band = false;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    nms_test(nb, area, bcast_lane(rx0, k + u), bcast_lane(ry0, k + u), bcast_lane(rx1, k + u),
                             bcast_lane(ry1, k + u), bcast_lane(ra, k + u), thr, hit, band);
                if (__ballot(band && !hit) != 0ull) {       // wave-uniform, practically never
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (nms_suppresses_exact(nb, area, bcast_lane(rx0, k + u), bcast_lane(ry0, k + u),
                                                 bcast_lane(rx1, k + u), bcast_lane(ry1, k + u), bcast_lane(ra, k + u), thr))
                            hit = true;
                }
            };
            int k = 0;
            for (; k + 4 <= kA; k += 4) group4(ax0, ay0, ax1, ay1, aa, k);
            for (; k < kA; ++k)
                if (nms_suppresses(nb, area, bcast_lane(ax0, k), bcast_lane(ay0, k), bcast_lane(ax1, k),
                                   bcast_lane(ay1, k), bcast_lane(aa, k), thr))
                    hit = true;
            k = WAVE;
            for (; k + 4 <= kB; k += 4) group4(bx0, by0, bx1, by1, ba, k - WAVE);
            for (; k < kB; ++k)
                if (nms_suppresses(nb, area, bcast_lane(bx0, k - WAVE), bcast_lane(by0, k - WAVE),
                                   bcast_lane(bx1, k - WAVE), bcast_lane(by1, k - WAVE), bcast_lane(ba, k - WAVE), thr))
                    hit = true;
            if (hit) alive = false;
        }
        for (int k = NMS_REG_KEPT; k < kept; ++k) {
            const float4 q = s_box[k];
            if (nms_suppresses(nb, area, q.x, q.y, q.z, q.w, s_area[k], thr)) alive = false;
        }
        // (2) inside the chunk: the lowest live lane is kept and suppresses the lanes above it
        uint64_t mask = __ballot(alive);
        while (mask != 0ull && kept < max_ret) {
            const int j = __ffsll((unsigned long long)mask) - 1;
            const float jx0 = bcast_lane(nb.x, j), jy0 = bcast_lane(nb.y, j), jx1 = bcast_lane(nb.z, j),
                        jy1 = bcast_lane(nb.w, j), ja = bcast_lane(area, j);
            const bool slot = (int)lane == (kept & 63);
            if (kept < WAVE) {
                ax0 = slot ? jx0 : ax0; ay0 = slot ? jy0 : ay0; ax1 = slot ? jx1 : ax1;
                ay1 = slot ? jy1 : ay1; aa = slot ? ja : aa;
            } else if (kept < NMS_REG_KEPT) {
                bx0 = slot ? jx0 : bx0; by0 = slot ? jy0 : by0; bx1 = slot ? jx1 : bx1;
                by1 = slot ? jy1 : by1; ba = slot ? ja : ba;
            }
            if ((int)lane == j) {
                if (kept >= NMS_REG_KEPT) {
                    s_box[kept] = nb;
                    s_area[kept] = area;
                }
                s_key[kept] = key;
            }
            ++kept;
            const bool hit = nms_suppresses(nb, area, jx0, jy0, jx1, jy1, ja, thr);
            if ((int)lane > j && hit) alive = false;
            mask = __ballot(alive) & ~((2ull << j) - 1ull);
        }
        sort_sync<false>();
    }

    // append this segment's kept keys to the image's final list (order fixed by the next sort)
    int out_base = 0;
    if (lane == 0) out_base = atomicAdd(fin_count + b * CNT_STRIDE, kept);
    out_base = __shfl(out_base, 0);
    for (int k = (int)lane; k < kept; k += WAVE) {
        const uint32_t pos = (uint32_t)(out_base + k);
        if (pos < fin_stride) put_key(fin_keys + (size_t)b * fin_stride + pos, cand_to_final(s_key[k]), done != nullptr);
    }
    return (int)b;
}

// ------------------------------------------------------------------------------------------
// Block-cooperative greedy NMS for the big segments (NMS_COOP_MIN < m <= NMS_SEG_LDS candidates
// of one (image, class)).  As one wave's serial chain such a segment is the critical path of the
// whole launch (a 325-candidate class: ~50 us), and the greedy order cannot be cut -- but the
// tests against the kept boxes can be spread.  Wave w of the block owns candidates 64w .. 64w+63
// of the sorted segment:
//   (1) the block rank-sorts the keys and decodes the m boxes (registers + LDS);
//   (2) chunks are resolved in order: the owner walks its live lanes ("keep the first, test the
//       lanes above it, clear the suppressed ones"), publishes the kept mask, and the waves of
//       ALL later chunks test their candidates against those kept boxes at the same time.
// The chain per chunk is one walk over the chunk's kept boxes plus one parallel test pass, instead
// of a walk over the whole kept list per chunk.  Measured alternatives: a full m x m suppression
// bit matrix (4-5x the tests saturate the SIMDs: 49 us) and per-chunk 64 x 64 bit tiles ahead of
// the walk (1728 instructions per chunk whether rows survive or not: 40 us).  Greedy order, the TF
// NonMaxSuppressionV3 test and the max_ret cut are those of the wave form: identical keep-list.
// ------------------------------------------------------------------------------------------
// LDS of a cooperative block: sorted keys, unsorted-then-kept keys, and TWO 64-box buffers -- only the boxes of the
// chunk just resolved are ever read by other waves, so the owner publishes its 64 boxes at the end of its walk
// (double buffered: the readers of chunk ch are done before the barrier that chunk ch+1 ends with).  10.9 KB
// instead of 18.5 KB with all 512 boxes in LDS: two blocks fit beside the resident score-scan waves of another stream.
constexpr size_t COOP_LDS = (size_t)NMS_SEG_LDS * (8 + 8) + 2 * WAVE * (16 + 4) + NMS_WAVES * 16 + 16;

template <typename ItemDone>
__device__ __forceinline__ void nms_coop_block(
    unsigned char *s_raw, uint32_t block, uint32_t nblocks, const float4 *__restrict__ anchors,
    const BoxSrc rb, int N, int C, int do_clip, float hi_x, float hi_y, NmsThr thr, int max_ret,
    const uint64_t *__restrict__ cand_keys, uint32_t cand_stride, const int *__restrict__ seg_start,
    const int *__restrict__ coop_count, const int *__restrict__ coop_list, uint32_t total_segments,
    int *__restrict__ fin_count, uint64_t *__restrict__ fin_keys, uint32_t fin_stride, bool stay, ItemDone item_done)
{
    static_assert(NMS_WAVES * WAVE == (int)NMS_SEG_LDS, "one wave per 64-candidate chunk");
    uint64_t *s_key = reinterpret_cast<uint64_t *>(s_raw);                         // sorted keys
    uint64_t *s_kkey = s_key + NMS_SEG_LDS;                                        // unsorted keys, then kept keys
    float4 *s_box = reinterpret_cast<float4 *>(s_kkey + NMS_SEG_LDS);              // [2][WAVE] boxes of the last two chunks
    float *s_area = reinterpret_cast<float *>(s_box + 2 * WAVE);                   // [2][WAVE]
    unsigned long long *s_keptmask = reinterpret_cast<unsigned long long *>(s_area + 2 * WAVE);
    int *s_nkept = reinterpret_cast<int *>(s_keptmask + NMS_WAVES);
    int *s_base = s_nkept + NMS_WAVES;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int items = min(*coop_count, (int)total_segments);
    for (int item = (int)block; item < items; item += (int)nblocks) {
        const uint32_t seg_id = min((uint32_t)coop_list[item], total_segments - 1u);
        const uint32_t b = seg_id / (uint32_t)C, c = seg_id - b * (uint32_t)C;
        const int start = seg_start[(size_t)b * (C + 1) + c];
        const uint32_t m = min((uint32_t)max(seg_start[(size_t)b * (C + 1) + c + 1] - start, 0), NMS_SEG_LDS);   // NMS_COOP_MIN < m <= NMS_SEG_LDS
        const uint32_t nw = (m + WAVE - 1) / WAVE;
        // waves without a chunk leave at once when this block has no further item: their slots go to
        // the per-wave blocks waiting behind (terminated waves do not count at the barriers)
        // (stay: pass 2, where the whole block may have to run an image's final top-k after the item)
        if (!stay && items <= (int)nblocks && wave >= nw) return;

        // (1) rank sort: keys are distinct (the anchor id is part of the key), so the number of smaller
        // keys is the sorted position -- m broadcast LDS reads per thread and one barrier instead of
        // the 45 barrier-separated stages of a 512-key bitonic network; then decode
        const uint64_t mine = tid < m ? cand_keys[(size_t)b * cand_stride + start + tid] : ~0ull;
        s_kkey[tid] = mine;
        __syncthreads();
        if (tid < m) {
            uint32_t r = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < m; ++j) r += s_kkey[j] < mine ? 1u : 0u;
            s_key[r] = mine;
        }
        __syncthreads();
        const uint32_t i = tid;                      // my candidate in visiting order
        const bool vi = i < m;
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
        float pa = 0.0f;
        if (vi) {
            const float4 v = candidate_box(anchors, rb, b, N, key_anchor(s_key[i]), do_clip, hi_x, hi_y);
            pb = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
            pa = (pb.w - pb.y) * (pb.z - pb.x);
        }

        // (2) chunks in order
        bool dead = !vi;
        int kept_total = 0;
        for (uint32_t ch = 0; ch < nw; ++ch) {
            if (wave == ch) {
                unsigned long long alive = __ballot(!dead);
                unsigned long long keptmask = 0ull;
                int kept = kept_total;
                while (alive != 0ull && kept < max_ret) {
                    const int j = __ffsll(alive) - 1;
                    keptmask |= 1ull << j;
                    ++kept;
                    const bool hit = nms_suppresses(pb, pa, bcast_lane(pb.x, j), bcast_lane(pb.y, j), bcast_lane(pb.z, j),
                                                    bcast_lane(pb.w, j), bcast_lane(pa, j), thr);
                    alive &= ~(__ballot(hit) | ((2ull << j) - 1ull));    // lanes <= j are decided
                }
                if ((keptmask >> lane) & 1ull)
                    s_kkey[kept_total + __popcll(keptmask & ((1ull << lane) - 1ull))] = s_key[i];
                if (lane == 0) {
                    s_keptmask[ch] = keptmask;
                    s_nkept[ch] = kept;
                }
                s_box[(ch & 1u) * WAVE + lane] = pb;         // the chunk's boxes, for the waves of the later chunks
                s_area[(ch & 1u) * WAVE + lane] = pa;
            }
            __syncthreads();
            kept_total = s_nkept[ch];
            if (kept_total >= max_ret) break;        // block-uniform
            if (wave > ch && wave < nw) {
                const unsigned long long km0 = s_keptmask[ch];
                const unsigned long long keptmask = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(km0 >> 32)) << 32) |
                                                    (unsigned int)__builtin_amdgcn_readfirstlane((int)km0);
                const float4 jb = s_box[(ch & 1u) * WAVE + lane];
                const float ja = s_area[(ch & 1u) * WAVE + lane];
                bool hit = false, band = false;
                for (unsigned long long km = keptmask; km != 0ull; km &= km - 1ull) {
                    const int j = __ffsll(km) - 1;
                    nms_test(pb, pa, bcast_lane(jb.x, j), bcast_lane(jb.y, j), bcast_lane(jb.z, j), bcast_lane(jb.w, j),
                             bcast_lane(ja, j), thr, hit, band);
                }
                if (__ballot(band && !hit) != 0ull)  // wave-uniform, practically never
                    for (unsigned long long km = keptmask; km != 0ull; km &= km - 1ull) {
                        const int j = __ffsll(km) - 1;
                        if (nms_suppresses_exact(pb, pa, bcast_lane(jb.x, j), bcast_lane(jb.y, j), bcast_lane(jb.z, j),
                                                 bcast_lane(jb.w, j), bcast_lane(ja, j), thr))
                            hit = true;
                    }
                dead = dead || hit;
            }
        }
        if (tid == 0) *s_base = atomicAdd(fin_count + b * CNT_STRIDE, kept_total);
        __syncthreads();
        const int out_base = *s_base;
        for (int k = (int)tid; k < kept_total; k += NMS_WAVES * WAVE) {
            const uint32_t pos = (uint32_t)(out_base + k);
            if (pos < fin_stride) put_key(fin_keys + (size_t)b * fin_stride + pos, cand_to_final(s_kkey[k]), stay);
        }
        __syncthreads();
        item_done(b);
    }
}

// kernel 4: per image, order the kept detections like tf.math.top_k over the class-ordered
// concat (bndbox.py:220-229), emit the first max_ret and zero-pad the rest (:233-244).
// Fast path (max_ret <= 128, list fits LDS): only the 128 smallest keys are ever needed, so
// every wave sorts 128-key chunks in registers, then chunk pairs are merged keeping the lower
// half (min(A[e], B[127-e]) is bitonic and holds the 128 smallest of both; one register merge
// sorts it) -- log2(chunks) rounds instead of a full sort of the list.
#ifndef ODBOX_FIN_THREADS
#define ODBOX_FIN_THREADS 512                 // = PART_THREADS = the NMS block: the three share blocks in pass 2
#endif
constexpr int FIN_THREADS = ODBOX_FIN_THREADS;
constexpr uint32_t FIN_CHUNK = 128;

// The FIN_CHUNK smallest of nchunks * FIN_CHUNK keys in LDS (~0 = none), sorted, into chunk 0; by the whole block
// (NWAVES waves), which has a barrier behind it before and after.  Every wave sorts 128-key chunks in registers,
// then chunk pairs are merged keeping the lower half: min(A[e], B[127-e]) is bitonic and holds the 128 smallest of
// both, one register merge sorts it -- log2(chunks) rounds instead of a full sort.
template <int NWAVES = FIN_THREADS / WAVE>
__device__ __forceinline__ void best128_of_chunks(uint64_t *s_keys, uint32_t nchunks, uint32_t tid)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = NWAVES;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        uint64_t k[2];
        uint64_t *ch = s_keys + c * FIN_CHUNK;
        k[0] = ch[lane * 2];
        k[1] = ch[lane * 2 + 1];
        wave_sort_regs<2>(k, lane);
        ch[lane * 2] = k[0];
        ch[lane * 2 + 1] = k[1];
    }
    __syncthreads();
    for (uint32_t stride = 1; stride < nchunks; stride <<= 1) {
        for (uint32_t c = wave * 2 * stride; c + stride < nchunks; c += nwaves * 2 * stride) {
            uint64_t *lo = s_keys + c * FIN_CHUNK;
            const uint64_t *hi = s_keys + (c + stride) * FIN_CHUNK;
            uint64_t k[2];
#pragma unroll
            for (uint32_t s0 = 0; s0 < 2; ++s0) {
                const uint32_t e = lane * 2 + s0;
                const uint64_t x = lo[e], y = hi[FIN_CHUNK - 1 - e];
                k[s0] = x < y ? x : y;
            }
            wave_sort_regs<2, true>(k, lane);
            lo[lane * 2] = k[0];
            lo[lane * 2 + 1] = k[1];
        }
        __syncthreads();
    }
}

// the image's detections out of its sorted final keys k[0 .. keep): decode the winners, zero-pad to max_ret
// (bndbox.py:233-244)
template <typename Keys>
__device__ __forceinline__ void emit_detections(
    Keys k, uint32_t keep, uint32_t b, const float4 *__restrict__ anchors, const BoxSrc rb, int N, int do_clip, float hi_x,
    float hi_y, int max_ret, float4 *__restrict__ out_boxes, float *__restrict__ out_scores, int32_t *__restrict__ out_labels,
    int32_t *__restrict__ out_keep)
{
    for (uint32_t r = threadIdx.x; r < (uint32_t)max_ret; r += FIN_THREADS) {
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
}

// (a device function of the whole block, FIN_THREADS threads: pass 2 runs it from the NMS kernel, in the block that
// finishes an image's last segment)
__device__ __forceinline__ void finalize_image(
    uint64_t *s_keys, uint32_t b, const float4 *__restrict__ anchors, const BoxSrc rb, int N, int do_clip, float hi_x,
    float hi_y, int max_ret, const int *fin_count, uint64_t *fin_keys,
    uint32_t fin_stride, uint32_t lds_keys, float4 *__restrict__ out_boxes, float *__restrict__ out_scores,
    int32_t *__restrict__ out_labels, int32_t *__restrict__ out_keep, int32_t *__restrict__ out_count)
{
    const uint32_t tid = threadIdx.x;
    Stamps stp;
    uint64_t *gk = fin_keys + (size_t)b * fin_stride;
    const uint32_t n = min((uint32_t)fin_count[b * CNT_STRIDE], fin_stride);
    const uint32_t keep = min(n, (uint32_t)max_ret);
    ODBOX_LAP(stp, 56, tid == 0);                             // count loaded
    auto emit = [&](auto k) {
        emit_detections(k, keep, b, anchors, rb, N, do_clip, hi_x, hi_y, max_ret, out_boxes, out_scores, out_labels, out_keep);
    };
    if (out_count && tid == 0) out_count[b] = (int32_t)keep;

    if ((uint32_t)max_ret <= FIN_CHUNK) {
        // the list goes through LDS in batches of lds_keys keys: the 128 best so far are carried in chunk 0
        uint32_t carry = 0;
        for (uint32_t base = 0; base < n || base == 0;) {
            const uint32_t take = min(n - base, lds_keys - carry);
            const uint32_t cnt = carry + take;
            const uint32_t nchunks = (cnt + FIN_CHUNK - 1) / FIN_CHUNK;
            for (uint32_t i = carry + tid; i < nchunks * FIN_CHUNK; i += FIN_THREADS) s_keys[i] = i < cnt ? gk[base + i - carry] : ~0ull;
            __syncthreads();
            ODBOX_LAP(stp, 57, tid == 0);                     // keys in LDS
            best128_of_chunks(s_keys, nchunks, tid);
            base += take;
            carry = FIN_CHUNK;
            ODBOX_LAP(stp, 58, tid == 0);                     // chunk sorts and merges
            if (base >= n) break;
        }
        emit(s_keys);
        ODBOX_LAP(stp, 59, tid == 0);                         // winners decoded and stored
    } else if (n <= TAIL_SORT_CAP) {
        for (uint32_t i = tid; i < n; i += FIN_THREADS) s_keys[i] = gk[i];
        __syncthreads();
        sort_u64<true>(s_keys, n, tid, FIN_THREADS);
        emit(s_keys);
    } else {
        sort_u64<true>(gk, n, tid, FIN_THREADS);
        emit(gk);
    }
}


// kernel 4.  Pass 1 also decides whether the tier was enough for its image -- it was the whole list (state 1, from the
// partition kernel) or kept at least max_ret boxes -- and, if not, starts pass 2 of that image right here: the same
// block partitions the image's whole list (partition_image), so pass 2 costs one more launch (the NMS kernel, whose
// blocks also run the final top-k of the images they finish), not three.
__global__ __launch_bounds__(FIN_THREADS) void finalize_kernel(
    const float4 *__restrict__ anchors, const BoxSrc rb, int N, int do_clip, float hi_x,
    float hi_y, int max_ret, int *fin_count, uint64_t *fin_keys,
    uint32_t fin_stride, uint32_t lds_keys, float4 *__restrict__ out_boxes, float *__restrict__ out_scores,
    int32_t *__restrict__ out_labels, int32_t *__restrict__ out_keep, int32_t *__restrict__ out_count, hdr_word *hdr,
    int pass, int *__restrict__ state,
    const int *__restrict__ cand_count, uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
    uint32_t cand_stride, int C, uint32_t pre_nms_topk, int *__restrict__ seg_start, int *__restrict__ coop_count,
    int *__restrict__ coop_list)
{
    static_assert(FIN_THREADS == PART_THREADS, "the final kernel's block also partitions");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_fin[];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (hdr && b == 0 && tid == 0) hdr[0] = hdr[0] + 1ull;    // the call's last kernel: next epoch (see wait_ready)
    bool again = false;                                       // block-uniform
    if (pass == 1) {
        const uint32_t n = min((uint32_t)fin_count[b * CNT_STRIDE], fin_stride);
        again = !(state[b] != 0 || n >= (uint32_t)max_ret);
    }
    finalize_image(reinterpret_cast<uint64_t *>(s_fin), b, anchors, rb, N, do_clip, hi_x, hi_y, max_ret, fin_count, fin_keys,
                   fin_stride, lds_keys, out_boxes, out_scores, out_labels, out_keep, out_count);
    if (pass != 1) return;
    __syncthreads();                                          // everybody has read state[b] and is done with the LDS keys
    if (tid == 0) state[b] = again ? 0 : 1;
    if (again && tid == 0) atomicAdd(coop_count + 1, 1);      // (coop_count: pass 2's; behind it the number of images pass 2 has to do)
    if (again)
        partition_image(s_fin, b, gridDim.x * (uint32_t)C, cand_count, keys_in, keys_out, cand_stride, C, pre_nms_topk,
                        seg_start, coop_count, coop_list, 2, 0u, state, fin_count);
}

// ------------------------------------------------------------------------------------------
// Pass 1 in ONE launch (round 3): tier selection, class partition, per-class greedy NMS and the final top-k of one
// image by one block, everything between the candidate list and the detections in LDS and registers.  As three
// kernels pass 1 cost 4 + 12 + 7 us plus two kernel boundaries on the configs[4] share, nearly all of it latency:
// global round trips between phases (keys out and in again, segment starts, kept keys through atomically reserved
// slots) and 5 760 one-segment waves that each pay a launch slot, a key load and a box gather for nine candidates.
// Here the tier (at most TT_CAP candidates) is laid out densely in class order, ONE CANDIDATE PER LANE:
//   * positions 64 w .. 64 w + 63 are window w, one window per wave (16 waves);
//   * every lane ranks its key inside its class segment (keys are distinct: the number of smaller keys of the
//     segment is the sorted place), so the array is in visiting order (class asc, score desc, anchor asc);
//   * all boxes are gathered at once, one per lane, and published in LDS;
//   * a window resolves ALL its class segments together: a lane's leader is the first live lane of its segment (a
//     find-first-set on the live mask under the lane's segment mask); the leader is kept, the other live lanes of
//     the segment are tested against its box (an LDS read at a per-lane index) and leave when they overlap it or
//     when max_ret boxes of their class have been kept before them (bndbox.py:172) -- as many trips as the longest
//     kept list among the window's segments, not their sum;
//   * a segment that continues from earlier windows (at most one per window: the layout is dense) is done in rounds,
//     chunk r of every such segment in round r: after each round the lanes of the later chunks are tested against
//     the boxes the chunk kept (LDS reads), then resolve in their own round.  With segments of nine candidates on
//     average there are two rounds; one class of 800 takes thirteen.  Same visiting order, same test, same cap as
//     nms_wave_segment / nms_coop_block: identical keep-lists.
// The kept lanes pack their keys, turned into final keys, at the front of the array and the block's 128 best are
// sorted out as in finalize_image.
// THE LADDER.  Any tier that keeps max_ret boxes gives the exact detections, and the block's cost grows with the
// tier, so the first attempt takes a quarter of the largest tier (factor 8: the best 2 max_ret candidates).  If that
// keeps too few, ONE more attempt follows, sized from the first one's survival rate (twice 1.5 max_ret / rate, at
// most the largest tier); an image whose rate says that even the largest tier will not hold max_ret survivors, whose
// second attempt fails too, whose tier does not fit (more than TT_CAP keys with the ties at its edge) or whose list the
// pre-NMS top-k cuts is partitioned for pass 2 right here, like the pass-1 final kernel does it.  state[b] records
// which attempt finished the image (0: pass 2).  On the configs[4] share: 18.7 us against 4 + 12 + 7 us and two
// boundaries, one call 207 -> 197 us on one box (tools/exp_nms_single.py --fused 0 / 1).
// ------------------------------------------------------------------------------------------
#ifndef ODBOX_TT_THREADS
#define ODBOX_TT_THREADS 1024
#endif
#ifndef ODBOX_TIER_FIRST_DIV
#define ODBOX_TIER_FIRST_DIV 4                // first rung of the fused block's tier ladder = largest tier / this
#endif
constexpr int TT_THREADS = ODBOX_TT_THREADS;                          // 1024: one window per wave; 512: two
constexpr uint32_t TT_CAP = 1024;
constexpr int TT_WPW = (int)TT_CAP / TT_THREADS;                      // windows per wave
constexpr int TT_REG = PART_REG * PART_THREADS / TT_THREADS;          // keys per thread: the lists partition_image holds in registers
constexpr int TT_WAVES = TT_THREADS / WAVE;
constexpr uint32_t TT_WINDOWS = TT_CAP / WAVE;
constexpr size_t TT_REGION = (size_t)TT_CAP * sizeof(float4);         // tier histogram / keys + class histogram / boxes / final keys
__host__ __device__ constexpr size_t tier_tail_lds(int C)             // region, segment starts [C + 1], kept masks [2][windows], scratch
{
    return TT_REGION + (((size_t)(C + 1) * sizeof(int) + 15) & ~(size_t)15) + 2 * TT_WINDOWS * 8 + 64 * sizeof(int);
}
static_assert(PART_THREADS == FIN_THREADS && TT_THREADS >= PART_THREADS && TT_WPW >= 1 && TT_WPW * TT_THREADS == (int)TT_CAP,
              "the block's first PART_THREADS threads also partition for pass 2");
static_assert(tier_tail_lds(90) <= TAIL_LDS_FIT, "the fused pass-1 block does not fit beside a scan");
static_assert((TIER_BUCKETS + 2) * sizeof(int) <= TT_REGION / 2, "tier histogram in the lower half of the region");

__global__ __launch_bounds__(TT_THREADS, 5) void tier_tail_kernel(
    const float4 *__restrict__ anchors, const BoxSrc rb, int N, int C, int do_clip, float hi_x, float hi_y, NmsThr thr,
    int max_ret, const int *__restrict__ cand_count, uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
    uint32_t cand_stride, uint32_t pre_nms_topk, uint32_t tier_k, uint32_t tier_first, int *__restrict__ seg_start,
    int *__restrict__ coop_count2, int *__restrict__ coop_list, int *__restrict__ state, int *__restrict__ fin_count, float4 *__restrict__ out_boxes,
    float *__restrict__ out_scores, int32_t *__restrict__ out_labels, int32_t *__restrict__ out_keep,
    int32_t *__restrict__ out_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_tt[];
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(s_tt);                          // [TT_CAP]   lower half of the region
    int *s_tier = reinterpret_cast<int *>(s_tt);                                    // [TIER_BUCKETS + 2], before the keys
    int *s_hist = reinterpret_cast<int *>(s_tt + TT_REGION / 2);                    // [copies][C], upper half, until the scatter
    float4 *s_box = reinterpret_cast<float4 *>(s_tt);                               // [TT_CAP]   the whole region, after the sort
    int *s_start = reinterpret_cast<int *>(s_tt + TT_REGION);                       // [C + 1]
    unsigned long long *s_keptmask = reinterpret_cast<unsigned long long *>(s_tt + TT_REGION + (((size_t)(C + 1) * sizeof(int) + 15) & ~(size_t)15));
    int *s_misc = reinterpret_cast<int *>(s_keptmask + 2 * TT_WINDOWS);             // [0..31] scan scratch, [32] rounds, [33] kept
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int copies = part_copies(C);
    const uint32_t cmax = (uint32_t)C - 1u;

    Stamps stp;
    const uint32_t M_all = min((uint32_t)cand_count[b * CNT_STRIDE], cand_stride);
    ODBOX_LAP(stp, 40, tid == 0);                             // count loaded
    // block-uniform: lists the pre-NMS top-k cuts, or longer than the block's registers, are pass 2's whole
    bool again = (pre_nms_topk != 0u && M_all > pre_nms_topk) || M_all > (uint32_t)(TT_REG * TT_THREADS);
    // The tier ladder: the first attempt takes a SMALL tier (tier_first candidates: a quarter of the largest); if it keeps
    // max_ret boxes they are the image's detections (any tier that does is exact).  If not, the second and last attempt's
    // size is read off the first one's survival rate, and an image whose rate says that even the largest tier will not
    // do goes to pass 2 at once.  Every test below is block-uniform.
    uint32_t tk = min(tier_first, tier_k);
    // a list within the largest tier: whole at once, final whatever it keeps -- a bounded cost instead of a bet on the
    // first rung (800-candidate lists, tools/exp_nms_overlap.py: piles that collapse 196 us a call against 204 through
    // the rung, piles that half overlap 221 against 236; only lists that barely overlap would have won the bet)
    if (M_all <= tier_k) tk = tier_k;
    int rung = 1;                                             // state[b]: 0 = left to pass 2, else the attempt that finished the image
    while (!again) {
        uint32_t M = M_all;
        bool whole = false;
        uint32_t total_kept = 0;
        const uint64_t *in = keys_in + (size_t)b * cand_stride;
        uint64_t kreg[TT_REG];
#pragma unroll
        for (int r = 0; r < TT_REG; ++r) {
            const uint32_t i = tid + (uint32_t)r * TT_THREADS;
            kreg[r] = i < M ? in[i] : ~0ull;
        }
        whole = M <= tk;
        ODBOX_LAP(stp, 41, tid == 0);                         // keys loaded
        // the class histogram (upper half of the region: the tier selection works in the lower half and nothing writes
        // that half again before the scatter, three barriers on) is zeroed here, under the tier selection's own barriers
        for (int c = tid; c < copies * C; c += TT_THREADS) s_hist[c] = 0;
        if (!whole) M = tier_select<TT_THREADS, TT_REG>(kreg, M, tk, s_tier, s_misc, tid);
        else __syncthreads();
        ODBOX_LAP(stp, 42, tid == 0);                         // tier selected
        again = M > TT_CAP;                                   // (a run of equal scores at the tier's edge)
        if (again) break;
        {
            // ---- class histogram, dense segment starts, scatter into LDS ------------------------------------
            int *my_hist = s_hist + (int)(tid & (uint32_t)(copies - 1)) * C;
#pragma unroll
            for (int r = 0; r < TT_REG; ++r)
                if (kreg[r] != ~0ull) atomicAdd(&my_hist[min(key_class(kreg[r]), cmax)], 1);
            __syncthreads();
            const int chunk = (C + TT_THREADS - 1) / TT_THREADS;
            const int c0 = min((int)tid * chunk, C), c1 = min(c0 + chunk, C);
            int local = 0;
            for (int c = c0; c < c1; ++c)
                for (int k = 0; k < copies; ++k) local += s_hist[k * C + c];
            const int incl = wave_incl_scan(local);
            if (lane == 63u) s_misc[wave] = incl;
            if (tid == 0) {
                s_misc[32] = 0;
                s_misc[33] = 0;
            }
            __syncthreads();
            int run = incl - local + sum_first<TT_WAVES>(s_misc, wave);
            for (int c = c0; c < c1; ++c) {
                s_start[c] = run;
                int cnt = 0;
                for (int k = 0; k < copies; ++k) {            // each copy becomes the scatter cursor of its lanes
                    const int n = s_hist[k * C + c];
                    s_hist[k * C + c] = run + cnt;
                    cnt += n;
                }
                run += cnt;
            }
            if (tid == 0) s_start[C] = (int)M;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < TT_REG; ++r)
                if (kreg[r] != ~0ull) {
                    const uint32_t at = (uint32_t)atomicAdd(&my_hist[min(key_class(kreg[r]), cmax)], 1);
                    if (at < TT_CAP) s_keys[at] = kreg[r];
                }
            __syncthreads();
            ODBOX_LAP(stp, 43, tid == 0);                     // class histogram, starts, scatter
        }
        // ---- one candidate per lane: position p = tid + TT_THREADS h in the class-ordered array --------------------
        uint64_t key[TT_WPW];
        uint32_t seg_s[TT_WPW], seg_e[TT_WPW], place[TT_WPW];
        bool valid[TT_WPW];
#pragma unroll
        for (int h = 0; h < TT_WPW; ++h) {
            const uint32_t p = tid + (uint32_t)h * TT_THREADS;
            valid[h] = p < M;
            key[h] = valid[h] ? s_keys[p] : ~0ull;
            const uint32_t c = min(key_class(key[h]), cmax);
            seg_s[h] = valid[h] ? min((uint32_t)s_start[c], M) : 0u;
            seg_e[h] = valid[h] ? min((uint32_t)s_start[c + 1u], M) : 0u;
            // (four reads in flight per trip; a read past the segment's end stays inside the array and is not counted)
            uint32_t r = 0;
            for (uint32_t j = seg_s[h]; j < seg_e[h]; j += 4u) {
                uint64_t kk[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) kk[u] = s_keys[min(j + u, TT_CAP - 1u)];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) r += (j + u < seg_e[h] && kk[u] < key[h]) ? 1u : 0u;
            }
            place[h] = min(seg_s[h] + r, TT_CAP - 1u);
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < TT_WPW; ++h)
            if (valid[h]) s_keys[place[h]] = key[h];
        __syncthreads();
        ODBOX_LAP(stp, 44, tid == 0);                         // rank sort
        float4 nb[TT_WPW];
        float area[TT_WPW];
        uint32_t cls[TT_WPW], q[TT_WPW], q_in[TT_WPW];
        unsigned long long segm[TT_WPW];
        int kcnt[TT_WPW];
        bool dead[TT_WPW], kept[TT_WPW];
#pragma unroll
        for (int h = 0; h < TT_WPW; ++h) {
            const uint32_t p = tid + (uint32_t)h * TT_THREADS;
            key[h] = valid[h] ? s_keys[p] : ~0ull;
            cls[h] = min(key_class(key[h]), cmax);
            nb[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            area[h] = 0.0f;
            if (valid[h]) {
                const float4 v = candidate_box(anchors, rb, b, N, key_anchor(key[h]), do_clip, hi_x, hi_y);
                nb[h] = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
                area[h] = (nb[h].w - nb[h].y) * (nb[h].z - nb[h].x);
            }
            // the segment that comes in from earlier windows, if any, holds lane 0 of the window; its chunk index
            // here is the number of windows back to its start
            const uint32_t w = wave + (uint32_t)h * TT_WAVES;
            seg_s[h] = valid[h] ? min((uint32_t)s_start[cls[h]], M) : 0u;
            seg_e[h] = valid[h] ? min((uint32_t)s_start[cls[h] + 1u], M) : 0u;
            {   // the lanes of this window that hold my class segment (contiguous: the array is in class order)
                const uint32_t w0 = w * WAVE;
                const uint32_t a = seg_s[h] > w0 ? seg_s[h] - w0 : 0u;
                const uint32_t e = seg_e[h] > w0 ? min(seg_e[h] - w0, (uint32_t)WAVE) : 0u;
                segm[h] = (e >= (uint32_t)WAVE ? ~0ull : (1ull << e) - 1ull) & ~((1ull << min(a, 63u)) - 1ull);
                if (!valid[h]) segm[h] = 0ull;
            }
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_s[h]);
            const uint32_t c_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)cls[h]);
            const bool has_in = (w * WAVE < M) && s0 < w * WAVE;            // wave-uniform
            q_in[h] = has_in ? w - (s0 >> 6) : 0u;
            q[h] = (has_in && cls[h] == c_first) ? q_in[h] : 0u;
            kcnt[h] = 0;
            dead[h] = !valid[h];
            kept[h] = false;
            if (lane == 0 && q_in[h] != 0u) atomicMax(&s_misc[32], (int)q_in[h]);
        }
        __syncthreads();                                      // every sorted key has been read: the region takes the boxes
#pragma unroll
        for (int h = 0; h < TT_WPW; ++h) s_box[tid + (uint32_t)h * TT_THREADS] = nb[h];
        __syncthreads();
        ODBOX_LAP(stp, 45, tid == 0);                         // boxes gathered and published
        const uint32_t rounds = (uint32_t)s_misc[32] + 1u;   // block-uniform
        for (uint32_t r = 0; r < rounds; ++r) {
            // every class segment of a window at once: a lane's LEADER is the first live lane of its segment; the leader
            // is kept, the others are tested against its box (an LDS read at a per-lane index) -- as many trips as the
            // longest kept list among the window's segments, not their sum.  Both windows of the wave share the loop.
            unsigned long long alive[TT_WPW], any = 0ull;
            bool lead_r[TT_WPW];
#pragma unroll
            for (int h = 0; h < TT_WPW; ++h) {
                lead_r[h] = false;
                alive[h] = __ballot(!dead[h] && q[h] == r);
                any |= alive[h];
            }
            while (any != 0ull) {
                any = 0ull;
#pragma unroll
                for (int h = 0; h < TT_WPW; ++h) {
                    if (alive[h] == 0ull) continue;           // wave-uniform
                    const uint32_t w = wave + (uint32_t)h * TT_WAVES;
                    const unsigned long long my = alive[h] & segm[h];
                    const bool act = ((alive[h] >> lane) & 1ull) != 0ull;
                    const uint32_t leader = my != 0ull ? (uint32_t)(__ffsll(my) - 1) : lane;
                    const float4 jb = s_box[w * WAVE + leader];
                    const float ja = (jb.w - jb.y) * (jb.z - jb.x);
                    bool hit = false, band = false;
                    nms_test(nb[h], area[h], jb.x, jb.y, jb.z, jb.w, ja, thr, hit, band);
                    if (__ballot(act && band && !hit) != 0ull)            // wave-uniform, practically never
                        if (nms_suppresses_exact(nb[h], area[h], jb.x, jb.y, jb.z, jb.w, ja, thr)) hit = true;
                    const bool lead = act && leader == lane;
                    kcnt[h] += act ? 1 : 0;                   // one more kept box of my class
                    lead_r[h] = lead_r[h] || lead;
                    alive[h] = __ballot(act && !lead && !hit && kcnt[h] < max_ret);
                    any |= alive[h];
                }
            }
#pragma unroll
            for (int h = 0; h < TT_WPW; ++h) {
                const uint32_t w = wave + (uint32_t)h * TT_WAVES;
                kept[h] = kept[h] || lead_r[h];
                const unsigned long long keptm = __ballot(lead_r[h]);
                if (lane == 0) s_keptmask[(r & 1u) * TT_WINDOWS + w] = keptm;
            }
            if (r + 1u >= rounds) break;
            __syncthreads();
#pragma unroll
            for (int h = 0; h < TT_WPW; ++h) {
                if (q_in[h] <= r) continue;                   // wave-uniform
                // chunk r of the segment that comes in lives in window src: its lanes there run from the segment's
                // start (r == 0) or from lane 0 to the window's end
                const uint32_t w = wave + (uint32_t)h * TT_WAVES;
                const uint32_t src = w - q_in[h] + r;
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_s[h]);
                const uint32_t first = s0 > src * WAVE ? s0 - src * WAVE : 0u;
                unsigned long long km = s_keptmask[(r & 1u) * TT_WINDOWS + src];
                km = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(km >> 32)) << 32) |
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)km);
                km &= ~((1ull << first) - 1ull);
                const bool mine = q[h] == q_in[h] && !dead[h];
                bool hit = false;
                for (; km != 0ull; km &= km - 1ull) {
                    const int j = __ffsll(km) - 1;
                    const float4 jb = s_box[src * WAVE + (uint32_t)j];
                    const float ja = (jb.w - jb.y) * (jb.z - jb.x);
                    if (nms_suppresses(nb[h], area[h], jb.x, jb.y, jb.z, jb.w, ja, thr)) hit = true;
                    kcnt[h] += q[h] == q_in[h] ? 1 : 0;
                }
                if (mine && (hit || kcnt[h] >= max_ret)) dead[h] = true;
            }
        }
        ODBOX_LAP(stp, 46, lane == 0);                        // rounds (per wave)
        // ---- final top-k: the kept lanes' keys in final order, packed (any order: they are sorted next) in place of
        // the boxes, so that the sort below sees ceil(kept / 128) chunks, not ceil(tier / 128)
        uint32_t slot[TT_WPW];
        {
            unsigned long long km[TT_WPW];
            uint32_t n0 = 0;
#pragma unroll
            for (int h = 0; h < TT_WPW; ++h) {
                km[h] = __ballot(kept[h]);
                slot[h] = n0 + (uint32_t)__popcll(km[h] & ((1ull << lane) - 1ull));
                n0 += (uint32_t)__popcll(km[h]);
            }
            int base = 0;
            if (lane == 0 && n0 != 0u) base = atomicAdd(&s_misc[33], (int)n0);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int h = 0; h < TT_WPW; ++h) slot[h] += (uint32_t)base;
        }
        __syncthreads();                                      // every box has been read, every wave has its slots
        total_kept = (uint32_t)s_misc[33];
        const uint32_t nchunks = max((total_kept + FIN_CHUNK - 1) / FIN_CHUNK, 1u);
        for (uint32_t p = total_kept + tid; p < nchunks * FIN_CHUNK; p += TT_THREADS) s_keys[p] = ~0ull;
#pragma unroll
        for (int h = 0; h < TT_WPW; ++h)
            if (kept[h] && slot[h] < TT_CAP) s_keys[slot[h]] = cand_to_final(key[h]);
        __syncthreads();
        ODBOX_LAP(stp, 47, tid == 0);                         // final keys in LDS (waits for the slowest wave)
        if (whole || total_kept >= (uint32_t)max_ret) {
            uint64_t *s_best = s_keys;                        // the sorted best FIN_CHUNK keys
            if (nchunks <= 4u && TT_THREADS == 1024) {
                // at most 512 kept keys (the usual first rung, every short list): a key's place is the number of smaller
                // keys -- the block's threads share the keys out, 1024 / K threads per key, each counts its part of the
                // K keys (broadcast reads) and adds it up in LDS.  The 128-key sorting network costs 2.1 us even for an
                // empty list (28 shuffle stages), this 0.5; from 1 024 keys on the network wins.
                const uint32_t K = nchunks <= 1u ? FIN_CHUNK : nchunks == 2u ? 2u * FIN_CHUNK : 4u * FIN_CHUNK;
                // keys each thread looks at: its share of the kept ones (in eights), at most 16 / 64 / 256
                const uint32_t tpk = (uint32_t)TT_THREADS / K;                       // threads per key
                const uint32_t per = min(K / tpk, ((total_kept + tpk - 1u) / tpk + 7u) & ~7u);
                int *s_rank = reinterpret_cast<int *>(s_keys + K);                   // [K], then the sorted best
                s_best = s_keys + K + K / 2u;
                for (uint32_t p = nchunks * FIN_CHUNK + tid; p < K; p += TT_THREADS) s_keys[p] = ~0ull;     // (three chunks: pad to four)
                if (tid < K) s_rank[tid] = 0;
                __syncthreads();
                const uint32_t t = tid & (K - 1u), part = tid / K;
                const uint64_t mine = s_keys[t];
                const uint64_t *from = s_keys + part * per;
                uint32_t r = 0;
                if (t < total_kept) {                                               // (the padding needs no place)
                    for (uint32_t j = 0; j < per; j += 8u) {
                        uint64_t kk[8];
#pragma unroll
                        for (uint32_t u = 0; u < 8u; ++u) kk[u] = from[j + u];
#pragma unroll
                        for (uint32_t u = 0; u < 8u; ++u) r += kk[u] < mine ? 1u : 0u;
                    }
                }
                if (r != 0u) atomicAdd(&s_rank[t], (int)r);
                __syncthreads();
                if (tid < total_kept && (uint32_t)s_rank[tid] < FIN_CHUNK) s_best[s_rank[tid]] = mine;
                __syncthreads();
            } else {
                best128_of_chunks<TT_WAVES>(s_keys, nchunks, tid);
            }
            ODBOX_LAP(stp, 48, tid == 0);                     // chunk sorts and merges
            const uint32_t keep = min(total_kept, (uint32_t)max_ret);
            if (out_count && tid == 0) out_count[b] = (int32_t)keep;
            emit_detections(s_best, keep, b, anchors, rb, N, do_clip, hi_x, hi_y, max_ret, out_boxes, out_scores, out_labels, out_keep);
            ODBOX_LAP(stp, 49, tid == 0);                     // winners decoded and stored
            break;
        }
        // not enough survivors.  ONE more attempt, if the survival rate says that a tier within the largest one can
        // hold max_ret of them (with half as many again to spare); it takes twice that estimate -- the rate falls with
        // depth, and a third attempt would cost more than the chip-wide pass 2 (tools/exp_nms_overlap.py)
        const uint64_t want = ((uint64_t)max_ret * M * 3u) / (2u * max(total_kept, 1u)) + 1u;
        const bool fits_whole = M_all <= tier_k;              // the largest tier is the whole list: final whatever it keeps
        again = rung >= 2 || tk >= tier_k || (want > (uint64_t)tier_k && !fits_whole);
        if (again) break;
        tk = fits_whole ? tier_k : (uint32_t)min((uint64_t)tier_k, max(2u * want, (uint64_t)2u * tk));
        ++rung;
        __syncthreads();                                      // the LDS starts over
    }
    if (tid == 0) state[b] = again ? 0 : rung;
    if (again && tid == 0) atomicAdd(coop_count2 + 1, 1);     // the number of images pass 2 has to do
    if (again) {
        __syncthreads();                                      // the LDS is free
        if (tid >= (uint32_t)PART_THREADS) return;            // (whole waves; waves that have ended do not count at a barrier)
        partition_image(s_tt, b, gridDim.x * (uint32_t)C, cand_count, keys_in, keys_out, cand_stride, C, pre_nms_topk,
                        seg_start, coop_count2, coop_list, 2, 0u, state, fin_count);
    }
}

// kernel 3: the first coop_blocks blocks take the big segments cooperatively (they start first, so
// the longest chains overlap everything else), the others run one wave per (image, class) segment.
struct FinalArgs {                     // what finalize_image needs beyond the NMS kernel's own arguments
    uint32_t lds_keys;
    float4 *out_boxes;
    float *out_scores;
    int32_t *out_labels, *out_keep, *out_count;
};

// fuse (pass 2 only): the block that finishes the LAST of an image's C segments also runs the image's final top-k, so
// pass 2 is this one launch.  Every block tells how many segments of an image it has finished with one atomic add on
// the image's counter (fin_count line, word 1; zeroed when pass 2 of the image was started) once the kept keys its
// waves stored -- agent-scope stores, written through the L2 (put_key) -- have been acknowledged; the block whose add
// completes the count acquires and reads them.  No block ever waits for another.  Pass 1 and the single plain pass keep their own final kernel and pay none of this.
template <bool FUSE>
__global__ __launch_bounds__(NMS_WAVES *WAVE) void nms_kernel(
    const float4 *__restrict__ anchors, const BoxSrc rb, int N, int C, int do_clip,
    float hi_x, float hi_y, NmsThr thr, int max_ret, const uint64_t *__restrict__ cand_keys,
    uint32_t cand_stride, const int *__restrict__ seg_start,
    const int *__restrict__ coop_count, const int *__restrict__ coop_list, uint32_t coop_blocks,
    int *fin_count, uint64_t *fin_keys, uint32_t fin_stride,
    uint32_t total_segments, int waves_per_block, uint32_t per_wave_bytes, uint32_t sort_cap, uint32_t kept_alias,
    const int *__restrict__ done, FinalArgs fa, hdr_word *hdr, uint32_t vblocks, uint32_t dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    ODBOX_NO_EXPERIMENTS(dbg);
    if constexpr (!FUSE) {
        // pass 1 and the single plain pass: exactly the kernel of round 2 (its own final kernel follows)
        if ((dbg & 4u) && blockIdx.x < coop_blocks) return;       // timing experiments: one path only
        if ((dbg & 8u) && blockIdx.x >= coop_blocks) return;
        if (blockIdx.x < coop_blocks)
            nms_coop_block(s_raw, blockIdx.x, coop_blocks, anchors, rb, N, C, do_clip, hi_x, hi_y, thr, max_ret, cand_keys,
                           cand_stride, seg_start, coop_count, coop_list, total_segments, fin_count, fin_keys, fin_stride, false,
                           [](uint32_t) {});
        else
            nms_wave_segment(s_raw, blockIdx.x - coop_blocks, anchors, rb, N, C, do_clip, hi_x, hi_y, thr, max_ret,
                             cand_keys, cand_stride, seg_start, fin_count, fin_keys, fin_stride, total_segments,
                             waves_per_block, per_wave_bytes, sort_cap, coop_blocks != 0, kept_alias != 0, done, dbg);
    } else {
        __shared__ int s_owner[NMS_WAVES];
        __shared__ int s_last;
        if (hdr && blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = hdr[0] + 1ull;      // the call's last kernel: next epoch (see wait_ready)
        // no image was left to pass 2 (the usual case): every block leaves on one scalar load
        if (coop_count[1] == 0) return;
        // called by every thread of the block with block-uniform arguments
        auto segments_done = [&](uint32_t b, int count) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's kept keys have left it
            __syncthreads();
            if (threadIdx.x == 0) {
                // (no release fence: the kept keys went out as agent-scope stores, see put_key, and every wave has waited
                // for its own before the barrier above)
                const int before = atomicAdd(fin_count + b * CNT_STRIDE + 1, count);
                const bool last = before + count >= C;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                s_last = last ? 1 : 0;
            }
            __syncthreads();
            if (s_last != 0)
                finalize_image(reinterpret_cast<uint64_t *>(s_raw), b, anchors, rb, N, do_clip, hi_x, hi_y, max_ret, fin_count,
                               fin_keys, fin_stride, fa.lds_keys, fa.out_boxes, fa.out_scores, fa.out_labels, fa.out_keep, fa.out_count);
            __syncthreads();                                       // s_last and the LDS are free again
        };
        // The launch is a few blocks per CU that walk the block ids of the pass-1 geometry (vblocks of them).
        for (uint32_t vb = blockIdx.x; vb < vblocks; vb += gridDim.x) {
            if (vb < coop_blocks) {
                nms_coop_block(s_raw, vb, coop_blocks, anchors, rb, N, C, do_clip, hi_x, hi_y, thr, max_ret, cand_keys,
                               cand_stride, seg_start, coop_count, coop_list, total_segments, fin_count, fin_keys, fin_stride, true,
                               [&](uint32_t b) { segments_done(b, 1); });
                continue;
            }
            const int owner = nms_wave_segment(s_raw, vb - coop_blocks, anchors, rb, N, C, do_clip, hi_x, hi_y, thr, max_ret,
                                               cand_keys, cand_stride, seg_start, fin_count, fin_keys, fin_stride, total_segments,
                                               waves_per_block, per_wave_bytes, sort_cap, coop_blocks != 0, kept_alias != 0, done, dbg);
            if ((threadIdx.x & 63u) == 0u) s_owner[threadIdx.x >> 6] = owner;
            __syncthreads();
            // the block's segments are consecutive, so the waves that finished segments of one image are neighbours
            for (int w = 0; w < waves_per_block;) {
                const int ob = s_owner[w];
                int w2 = w;
                while (w2 < waves_per_block && s_owner[w2] == ob) ++w2;
                if (ob >= 0) segments_done((uint32_t)ob, w2 - w);
                w = w2;
            }
            __syncthreads();                                       // s_owner is free again
        }
    }
}

// ------------------------------------------------------------------------------------------
// workspace layout + launch geometry of odbox_decode_nms
// ------------------------------------------------------------------------------------------
// Tuning and experiment hooks (ODBOX_SCAN_*, ODBOX_NMS_DBG; used by tools/exp_*.sh) are read from
// the environment only in a build made with ODBOX_TUNING_HOOKS=1 (object_detection_amd/build.py);
// some of them switch parts of a kernel off, so a product build must not look at them.
static inline const char *tuning_env(const char *name)
{
#ifdef ODBOX_TUNING_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

struct NmsLayout {
    size_t off_counts, off_tickets, off_seg, off_coop, off_keys_a, off_keys_b, off_fin, total;
    uint32_t cand_stride, fin_stride;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool nms_layout(int B, int N, int C, int max_ret, NmsLayout &L)
{
    if (B < 1 || N < 1 || C < 1 || max_ret < 1) return false;
    if (N > ODBOX_MAX_ANCHORS || C > ODBOX_MAX_CLASSES || max_ret > ODBOX_MAX_RET) return false;
    L.cand_stride = (uint32_t)N;
    uint64_t fin = (uint64_t)C * (uint64_t)max_ret;
    if (fin > (uint64_t)N) fin = (uint64_t)N;
    L.fin_stride = (uint32_t)fin;
    L.off_counts = HDR_BYTES;                                      // epoch / ready line; then cand_count, fin_count (one 256-byte line per image), B spare ints; then the tile tickets
    L.off_tickets = L.off_counts + ((size_t)2 * B * CNT_STRIDE + align_up((size_t)B, 64) + CNT_STRIDE) * sizeof(int);   // ... spare[B], coop_count
    L.off_seg = L.off_tickets + (size_t)SCAN_QUEUES * SCAN_TICKET_STRIDE * sizeof(int);
    L.off_coop = L.off_seg + align_up((size_t)B * (C + 1) * sizeof(int), 256);                      // coop_list[B*C]
    L.off_keys_a = L.off_coop + align_up((size_t)B * C * sizeof(int), 256);
    L.off_keys_b = L.off_keys_a + align_up((size_t)B * L.cand_stride * sizeof(uint64_t), 256);
    L.off_fin = L.off_keys_b + align_up((size_t)B * L.cand_stride * sizeof(uint64_t), 256);
    L.total = L.off_fin + align_up((size_t)B * L.fin_stride * sizeof(uint64_t), 256);
    return true;
}

// worst number of lanes of one LDS lane group that land on one bank with different addresses,
// for lane l reading V dwords at (l >> LL) * C + (l & (LPR-1)) * V  (MI355X_MICROARCH.md, LDS)
int scan_bank_conflict(int C, int V, int LL)
{
    static const int g128[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int nbanks = V == 1 ? 32 : 64, lpr = 1 << LL;
    const int ngroups = V == 4 ? 4 : 2, glen = V == 4 ? 16 : 32;
    int worst = 1;
    for (int gi = 0; gi < ngroups; ++gi) {
        int hits[64] = {0};
        for (int i = 0; i < glen; ++i) {
            const int l = V == 4 ? g128[gi][i] : gi * 32 + i;
            const long addr = (long)(l >> LL) * C + (long)(l & (lpr - 1)) * V;
            for (int e = 0; e < V; ++e) {
                const int bank = (int)((addr + e) % nbanks);
                if (++hits[bank] > worst) worst = hits[bank];
            }
        }
    }
    return worst;
}

// step_cap / tile_target: bytes of one wave-step at most, bytes of one tile about.  fp32 scores: 16 KB / 12 KB.
// 16-bit scores: 8 KB / 6 KB -- their rows are half as long, and with 64 rows per step (one lane per row) a tile
// of C = 90 was 11.5 KB again, 6 waves per CU; 32 rows per step and 6 KB tiles keep 11 waves per CU in flight
// (same-box A/B at config 5: 99.8 -> 95.1 us; the fp32 scan does not care: 175.4 / 175.6 / 176.9 us).
ScanGeom scan_geometry(int C, int N, size_t step_cap = 16 * 1024, size_t tile_target = 12 * 1024)
{
    ScanGeom g{};
    g.V = (C % 4 == 0) ? 4 : (C % 2 == 0) ? 2 : 1;
    const int nvec = C / g.V;
    // Lanes per row: as FEW as possible (fewest instructions per row: no cross-lane reduction,
    // no per-step overhead) while the LDS reads stay at most 2-way bank conflicted, the lanes
    // stay >= 85 % busy and one wave-step (64 / LPR rows) stays <= 16 KiB of LDS.
    int pick = -1;
    for (int l = 0; l <= 6 && pick < 0; ++l) {
        const int lpr = 1 << l, K = (nvec + lpr - 1) / lpr;
        const double eff = (double)nvec / ((double)K * lpr);
        const size_t step_bytes = (size_t)(WAVE >> l) * C * sizeof(float);
        if (eff >= 0.85 && scan_bank_conflict(C, g.V, l) <= 2 && step_bytes <= step_cap) pick = l;
    }
    if (pick < 0) {                    // very wide rows: keep the step small, take the best lane use
        double best = -1.0;
        for (int l = 0; l <= 6; ++l) {
            const int lpr = 1 << l, K = (nvec + lpr - 1) / lpr;
            const double score = (double)nvec / ((double)K * lpr) / scan_bank_conflict(C, g.V, l);
            const size_t step_bytes = (size_t)(WAVE >> l) * C * sizeof(float);
            if ((step_bytes <= step_cap || l == 6) && score > best) { best = score; pick = l; }
        }
    }
    g.lpr_log2 = pick;
    if (const char *e = tuning_env("ODBOX_SCAN_LPR_LOG2")) g.lpr_log2 = atoi(e) >= 0 && atoi(e) <= 6 ? atoi(e) : g.lpr_log2;
    const int lpr = 1 << g.lpr_log2;
    g.K = (nvec + lpr - 1) / lpr;
    g.kfull = nvec / lpr;
    const int rows_per_step = WAVE / lpr;
    const size_t step_bytes = (size_t)rows_per_step * C * sizeof(float);
    int steps = (int)(tile_target / step_bytes);                       // ~12 KiB per tile (6 KiB for 16-bit scores)
    const int max_steps = (int)SCAN_MAX_ROWS / rows_per_step;
    g.steps = steps < 1 ? 1 : (steps > max_steps ? max_steps : steps);
    if (const char *e = tuning_env("ODBOX_SCAN_STEPS")) g.steps = atoi(e) > 0 ? atoi(e) : g.steps;
    if (g.steps > max_steps) g.steps = max_steps;
    g.buf_floats = (uint32_t)align_up((size_t)g.steps * rows_per_step * C, 4);
    const uint32_t R = (uint32_t)g.steps * rows_per_step;
    const uint32_t tpi = ((uint32_t)N + R - 1) / R;
    g.tpi_magic = tpi == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / tpi);
    return g;
}

int device_cu_count()
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

// dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel
// (the attribute is a maximum and sticks to the function on that device: asked once per (device,
// kernel) and size, not on every call -- it was up to four runtime calls on the hot host path)
int allow_lds_impl(const void *kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(dev, kernel)];
    if (have >= bytes) return 0;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return 0;
}
template <typename K>
int allow_lds(K kernel, size_t bytes)
{
    return allow_lds_impl(reinterpret_cast<const void *>(kernel), bytes);
}

template <int V, int LL, bool HALF>
int launch_scan_inst(hdr_word *hdr, int *coop_count, const float *scores, int B, int N, int C, const ScanGeom &g, float score_thr, int *tile_ticket,
                     int *cand_count, uint64_t *cand_keys, uint32_t cand_stride, uint32_t half_mode, hipStream_t st)
{
    const uint32_t R = (uint32_t)g.steps * (WAVE >> LL);
    const uint64_t tiles = ((uint64_t)N + R - 1) / R;
    const uint64_t total = tiles * (uint64_t)B;
    if (total > 0x7fffffffull) return fail(ODBOX_EINVAL, "odbox_decode_nms: too many score tiles");
    const size_t lds = (size_t)2 * g.buf_floats * sizeof(float) + (size_t)R * 8;
    if (lds > 160 * 1024)
        return fail(ODBOX_EINVAL, "odbox_decode_nms: C=%d needs %zu bytes of LDS per wave", C, lds);
    int per_cu = (int)((160 * 1024) / align_up(lds, 1024));
    per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
    // leave TAIL_LDS_RESERVE bytes of every CU's LDS to other kernels: the partition / NMS / finalize
    // launches of ANOTHER batch (a second stream) then run beside this scan instead of behind it.
    // Measured at config 5: the scan is as fast with 5 resident waves per CU as with 6.
    while (per_cu > 2 && (size_t)per_cu * align_up(lds, 1024) + TAIL_LDS_RESERVE > 160 * 1024) --per_cu;
    // ... and no more than that: the allocation of each wave is padded until what is left of the CU's LDS is too
    // small for a scan wave of another stream.  Two scans side by side share the HBM rate, finish together and
    // leave the memory system idle while both streams run their tail kernels (2 streams: 209 us per batch);
    // with one scan at a time the other stream's tail kernels always have a scan to hide under.
    size_t lds_alloc = lds;
    if (!tuning_env("ODBOX_SCAN_NO_PAD")) {
        const size_t pad_to = ((160 * 1024 - TAIL_LDS_RESERVE) / per_cu) & ~(size_t)1023;
        if (pad_to > lds_alloc) lds_alloc = pad_to;
    }
    if (const char *e = tuning_env("ODBOX_SCAN_WAVES_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;   // tuning aid
    uint32_t nq = SCAN_QUEUES, dbg = 0;
    if (const char *e = tuning_env("ODBOX_SCAN_QUEUES")) nq = atoi(e) > 0 && atoi(e) <= (int)SCAN_QUEUES ? atoi(e) : nq;
    if (const char *e = tuning_env("ODBOX_SCAN_DBG")) dbg = (uint32_t)atoi(e);
    const uint64_t resident = (uint64_t)device_cu_count() * per_cu;
    const uint32_t grid = (uint32_t)(total < resident ? total : resident);
    if (int rc = allow_lds(score_scan_kernel<V, LL, HALF>, lds_alloc)) return rc;
    hipLaunchKernelGGL((score_scan_kernel<V, LL, HALF>), dim3(grid), dim3(WAVE), lds_alloc, st, hdr, (uint32_t)B, coop_count, scores, N, C, g, score_thr,
                       (uint32_t)tiles, (uint32_t)total, tile_ticket, cand_count, cand_keys, cand_stride, nq, half_mode, dbg);
    return launch_status("score_scan");
}

// (C = dwords per row: classes for fp32 scores, classes / 2 for 16-bit ones)
template <int V, bool HALF>
int launch_scan(hdr_word *hdr, int *coop_count, const float *scores, int B, int N, int C, const ScanGeom &g, float score_thr, int *tile_ticket,
                int *cand_count, uint64_t *cand_keys, uint32_t cand_stride, uint32_t half_mode, hipStream_t st)
{
#define ODBOX_SCAN_CASE(LL)                                                                                  \
    case LL:                                                                                                 \
        return launch_scan_inst<V, LL, HALF>(hdr, coop_count, scores, B, N, C, g, score_thr, tile_ticket, cand_count, cand_keys,    \
                                             cand_stride, half_mode, st)
    switch (g.lpr_log2) {
        ODBOX_SCAN_CASE(0);
        ODBOX_SCAN_CASE(1);
        ODBOX_SCAN_CASE(2);
        ODBOX_SCAN_CASE(3);
        ODBOX_SCAN_CASE(4);
        ODBOX_SCAN_CASE(5);
        ODBOX_SCAN_CASE(6);
    }
#undef ODBOX_SCAN_CASE
    return fail(ODBOX_EINVAL, "odbox_decode_nms: bad scan geometry");
}

}  // namespace

#ifdef ODBOX_STAMPS
extern "C" int odbox_debug_stamps(unsigned long long *sum, unsigned long long *mx, unsigned long long *cnt, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(sum, HIP_SYMBOL(g_stamp_sum), sizeof(g_stamp_sum)));
    HIP_TRY(hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_stamp_max), sizeof(g_stamp_max)));
    HIP_TRY(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_stamp_cnt), sizeof(g_stamp_cnt)));
    if (reset) {
        static const unsigned long long zero[64] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), zero, sizeof(zero)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_max), zero, sizeof(zero)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cnt), zero, sizeof(zero)));
    }
    return 0;
}
#endif

extern "C" {

int odbox_set_nms_tier(int factor)
{
    if (factor < 0 || factor > 4096) return fail(ODBOX_EINVAL, "odbox_set_nms_tier: factor %d (0 = off, at most 4096)", factor);
    return g_tier_factor.exchange(factor, std::memory_order_relaxed);
}

int odbox_set_nms_fused_tail(int on)
{
    if (on != 0 && on != 1) return fail(ODBOX_EINVAL, "odbox_set_nms_fused_tail: %d (0 or 1)", on);
    return g_fused_tail.exchange(on, std::memory_order_relaxed);
}

size_t odbox_decode_nms_workspace(int B, int N, int C, int max_ret, int pre_nms_topk)
{
    (void)pre_nms_topk;
    NmsLayout L;
    if (!nms_layout(B, N, C, max_ret, L)) return 0;
    return L.total;
}

}  // extern "C"

namespace {

bool dtype_ok(int d) { return d == ODBOX_DTYPE_F32 || d == ODBOX_DTYPE_F16 || d == ODBOX_DTYPE_BF16; }

// Body of odbox_decode_nms (fp32 model outputs: box_dtype = score_dtype = ODBOX_DTYPE_F32) and of
// odbox_decode_nms_mixed (16-bit ones).
int decode_nms_impl(const char *who, const float *anchors, const void *regress_or_boxes, int box_dtype, const void *scores,
                    int score_dtype, int B, int N, int C, float img_h, float img_w, int do_clip, float score_thr,
                    float iou_thr, int max_ret, int pre_nms_topk, float *out_boxes, float *out_scores, int32_t *out_labels,
                    int32_t *out_keep_anchor, int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream)
{
    if (B == 0) return 0;
    if (!regress_or_boxes || !scores || !out_boxes || !out_scores || !out_labels)
        return fail(ODBOX_EINVAL, "%s: NULL pointer", who);
    if (!dtype_ok(box_dtype) || !dtype_ok(score_dtype))
        return fail(ODBOX_EINVAL, "%s: dtypes must be ODBOX_DTYPE_F32, _F16 or _BF16", who);
    NmsLayout L;
    if (!nms_layout(B, N, C, max_ret, L))
        return fail(ODBOX_EINVAL,
                    "%s: bad shape B=%d N=%d C=%d max_ret=%d (limits: N<=%d C<=%d max_ret<=%d)", who, B, N, C,
                    max_ret, ODBOX_MAX_ANCHORS, ODBOX_MAX_CLASSES, ODBOX_MAX_RET);
    const bool half_scores = score_dtype != ODBOX_DTYPE_F32;
    if (half_scores && (C & 1))
        return fail(ODBOX_EINVAL, "%s: 16-bit scores are read two classes per dword: C=%d must be even (pad the class axis "
                                  "with a column that never exceeds the threshold)", who, C);
    if (pre_nms_topk < 0) return fail(ODBOX_EINVAL, "%s: pre_nms_topk < 0", who);
    if (!workspace || workspace_bytes < L.total)
        return fail(ODBOX_ENOSPC, "%s: workspace %zu bytes, need %zu", who, workspace_bytes, L.total);
    if (((uintptr_t)workspace & 255u) != 0) return fail(ODBOX_EINVAL, "%s: workspace must be 256-byte aligned", who);
    const uintptr_t box_align = box_dtype == ODBOX_DTYPE_F32 ? 15u : 7u;
    if ((anchors && !aligned16(anchors)) || ((uintptr_t)regress_or_boxes & box_align) || !aligned16(out_boxes) ||
        ((uintptr_t)scores & 3u))
        return fail(ODBOX_EINVAL, "%s: fp32 boxes must be 16-byte aligned (16-bit ones 8-byte), scores 4-byte aligned", who);

    hipStream_t st = (hipStream_t)stream;
    unsigned char *ws = (unsigned char *)workspace;
    int *cand_count = (int *)(ws + L.off_counts);
    int *fin_count = cand_count + (size_t)B * CNT_STRIDE;
    int *coop_count = cand_count + (size_t)2 * B * CNT_STRIDE + align_up((size_t)B, 64);
    int *coop_list = (int *)(ws + L.off_coop);
    int *tile_ticket = (int *)(ws + L.off_tickets);
    int *seg_start = (int *)(ws + L.off_seg);
    uint64_t *keys_a = (uint64_t *)(ws + L.off_keys_a);
    uint64_t *keys_b = (uint64_t *)(ws + L.off_keys_b);
    uint64_t *fin_keys = (uint64_t *)(ws + L.off_fin);
    const float hi_x = img_w - 1.0f, hi_y = img_h - 1.0f;
    const BoxSrc rb = {regress_or_boxes, box_dtype};

    hdr_word *hdr = (hdr_word *)ws;                                // epoch, ready: the scan zeroes the counters itself

    // kernel 1: score scan -> unordered candidate keys per image
    // (16-bit scores: two classes per dword, so the scan sees a [B*N, C/2] matrix of dwords)
    const int Cd = half_scores ? C / 2 : C;
    ScanGeom g = half_scores ? scan_geometry(Cd, N, 8 * 1024, 6 * 1024) : scan_geometry(Cd, N);
    g.stream = (uint64_t)B * (uint64_t)N * (uint64_t)Cd * sizeof(float) > (512ull << 20) ? 1u : 0u;
    if (const char *e = tuning_env("ODBOX_SCAN_STREAM")) g.stream = atoi(e) != 0;        // tuning aid
    const float *sp = (const float *)scores;
    int rc;
    if (half_scores) {
        // the packed integer maximum of the raw bits is the row maximum whenever that maximum is positive, which is
        // all a threshold >= 0 asks about; a negative threshold takes the order-mapped form (see the kernel)
        const uint32_t hm = (score_dtype == ODBOX_DTYPE_BF16 ? HALF_BF16 : 0u) | (score_thr >= 0.0f ? 0u : HALF_GENERAL);
        rc = g.V == 4 ? launch_scan<4, true>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, hm, st)
           : g.V == 2 ? launch_scan<2, true>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, hm, st)
                      : launch_scan<1, true>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, hm, st);
    } else {
        rc = g.V == 4 ? launch_scan<4, false>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, 0u, st)
           : g.V == 2 ? launch_scan<2, false>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, 0u, st)
                      : launch_scan<1, false>(hdr, coop_count, sp, B, N, Cd, g, score_thr, tile_ticket, cand_count, keys_a, L.cand_stride, 0u, st);
    }
    if (rc) return rc;

    // kernels 2-4, twice: pass 1 on every image's score tier, pass 2 on the whole list of the images whose tier did not
    // keep max_ret boxes (see cand_partition_kernel); pass 2 is left out when no list can be longer than the tier
    int *state = cand_count + (size_t)2 * B * CNT_STRIDE;          // [B]: 1 = pass 1 finished the image
    const int tier_factor = g_tier_factor.load(std::memory_order_relaxed);      // odbox_set_nms_tier; 0 = no tier
    const uint64_t tier_want = (uint64_t)tier_factor * (uint64_t)max_ret;
    const uint32_t tier_k = tier_factor > 0 && tier_want < 0xffffffffull ? (uint32_t)tier_want : 0xffffffffu;
    int passes = tier_k >= L.cand_stride ? 1 : 2;
    // hist + scan scratch, then room to sort one oversized class segment (only lists that can hold one need it) or
    // for the tier selection's score histogram
    size_t part_sort = L.cand_stride > NMS_SEG_LDS ? (size_t)(L.cand_stride < PART_SORT_CAP ? L.cand_stride : PART_SORT_CAP) * 8 : 0;
    if (part_sort < (TIER_BUCKETS + 2) * sizeof(int)) part_sort = (TIER_BUCKETS + 2) * sizeof(int);
    const size_t part_lds = part_sort_offset(C) + part_sort;
    if (int r2 = allow_lds(cand_partition_kernel, part_lds)) return r2;
    const uint64_t *nms_keys = keys_b;

    const size_t nms_extra = max_ret > NMS_REG_KEPT ? (size_t)(max_ret - NMS_REG_KEPT) : 0;
    // with the cooperative blocks on, a wave only ever sorts <= NMS_COOP_MIN keys itself
    const size_t kept_bytes = (size_t)((max_ret + 1) & ~1) * 8 + nms_extra * 20;
    size_t per_wave = align_up((size_t)NMS_COOP_MIN * 8 + kept_bytes, 16);
    uint32_t sort_cap = NMS_COOP_MIN;
    const int waves = per_wave * NMS_WAVES <= 64 * 1024 ? NMS_WAVES : 1;
    if (waves != NMS_WAVES) {                                  // no cooperative blocks: up to NMS_SEG_LDS keys per wave
        sort_cap = NMS_SEG_LDS;
        per_wave = align_up((size_t)NMS_SEG_LDS * 8 + kept_bytes, 16);
    }
    const uint64_t segs = (uint64_t)B * C;
    const NmsThr nms_thr = {iou_thr, iou_thr * (1.0f - 0x1p-20f), iou_thr * (1.0f + 0x1p-20f)};
    // cooperative blocks need the full block
    const uint32_t coop_blocks = waves == NMS_WAVES ? (uint32_t)(segs < ODBOX_COOP_BLOCKS ? segs : ODBOX_COOP_BLOCKS) : 0u;
    uint32_t kept_alias = 0;
#ifndef ODBOX_TUNING_HOOKS                                         // (the experiment switches can route a sorted segment to the general path)
    if (coop_blocks) {                                             // sort buffer and kept keys are never live together: share
        kept_alias = 1;
        per_wave = align_up(kept_bytes > (size_t)NMS_COOP_MIN * 8 ? kept_bytes : (size_t)NMS_COOP_MIN * 8, 16);
    }
#endif
    size_t nms_lds = per_wave * waves;
    if (coop_blocks && nms_lds < COOP_LDS) nms_lds = COOP_LDS;
    const uint64_t nms_grid = (segs + waves - 1) / waves + coop_blocks;
    if (int r3 = allow_lds(nms_kernel<false>, nms_lds)) return r3;

    // keys per LDS batch of the final kernel: a multiple of the 128-key chunk, at least two chunks (one carried + one new)
    size_t fin_keys_lds = align_up((size_t)(L.fin_stride < TAIL_SORT_CAP ? L.fin_stride : TAIL_SORT_CAP), FIN_CHUNK);
    if (fin_keys_lds < 2 * FIN_CHUNK) fin_keys_lds = 2 * FIN_CHUNK;
    const size_t fin_lds = fin_keys_lds * 8;
    if (waves != NMS_WAVES) passes = 1;                            // pass 2 runs the final top-k inside full NMS blocks
    const size_t fin_part_lds = passes == 2 && part_lds > fin_lds ? part_lds : fin_lds;      // pass 1: final top-k, then partition_image
    const size_t nms2_lds = nms_lds > fin_lds ? nms_lds : fin_lds;
    if (int r4 = allow_lds(finalize_kernel, fin_part_lds)) return r4;
    if (int r5 = allow_lds(nms_kernel<true>, nms2_lds)) return r5;

    // pass 2: four blocks per CU walk the pass-1 block ids (see nms_kernel<true>; a batch whose every image falls through:
    // 628 / 336 / 270 us with 64 / 256 / 1024 blocks, 280 us with one block per id; the empty launch costs the same)
    const uint32_t nms2_cap = 4u * (uint32_t)device_cu_count();
    const uint32_t nms2_grid = nms_grid < nms2_cap ? (uint32_t)nms_grid : nms2_cap;
    const FinalArgs fa = {(uint32_t)fin_keys_lds, (float4 *)out_boxes, out_scores, out_labels, out_keep_anchor, out_count};
    const uint32_t nms_dbg = (uint32_t)(tuning_env("ODBOX_NMS_DBG") ? atoi(tuning_env("ODBOX_NMS_DBG")) : 0);
    const int kpass = passes == 1 ? 0 : 1;                         // 0: one plain pass (no tier, no per-image state)
    // Pass 1 as ONE launch (tier_tail_kernel): when there is a tier, it leaves room in the block's TT_CAP lanes for the
    // ties at its edge, the final top-k is the 128-key form and the class histogram fits beside the keys
    const bool fused = g_fused_tail.load(std::memory_order_relaxed) != 0 && passes == 2 && max_ret <= (int)FIN_CHUNK &&
                       (uint64_t)tier_k + tier_k / 8 <= TT_CAP && (size_t)part_copies(C) * C * sizeof(int) <= TT_REGION / 2;
    if (fused) {
        // the ladder's first rung: a quarter of the largest tier (factor 8: the best 2 * max_ret candidates)
        const uint32_t tier_first = tier_factor >= 2 * ODBOX_TIER_FIRST_DIV ? tier_k / ODBOX_TIER_FIRST_DIV : tier_k;
        const size_t tt_lds = tier_tail_lds(C) > part_lds ? tier_tail_lds(C) : part_lds;      // (the block also partitions for pass 2)
        if (int r6 = allow_lds(tier_tail_kernel, tt_lds)) return r6;
        hipLaunchKernelGGL(tier_tail_kernel, dim3(B), dim3(TT_THREADS), tt_lds, st, (const float4 *)anchors, rb, N, C, do_clip,
                           hi_x, hi_y, nms_thr, max_ret, (const int *)cand_count, keys_a, keys_b, L.cand_stride,
                           (uint32_t)pre_nms_topk, tier_k, tier_first, seg_start, coop_count + 1, coop_list, state, fin_count,
                           (float4 *)out_boxes, out_scores, out_labels, out_keep_anchor, out_count);
        if (int r6 = launch_status("tier_tail")) return r6;
        hipLaunchKernelGGL(nms_kernel<true>, dim3(nms2_grid), dim3(waves * WAVE), nms2_lds, st, (const float4 *)anchors,
                           rb, N, C, do_clip, hi_x, hi_y, nms_thr, max_ret, nms_keys,
                           L.cand_stride, seg_start, (const int *)(coop_count + 1), (const int *)coop_list, coop_blocks,
                           fin_count, fin_keys, L.fin_stride, (uint32_t)segs, waves, (uint32_t)per_wave, sort_cap, kept_alias,
                           (const int *)state, fa, hdr, (uint32_t)nms_grid, nms_dbg);
        return launch_status("nms (pass 2)");
    }
    // kernel 2: (pass 1) the tier; (plain pass) the pre-NMS top-k where it binds; then group by class (counting sort)
    hipLaunchKernelGGL(cand_partition_kernel, dim3(B), dim3(PART_THREADS), part_lds, st, cand_count, keys_a, keys_b,
                       L.cand_stride, C, (uint32_t)pre_nms_topk, seg_start, coop_count, coop_list, kpass, tier_k, state, fin_count);
    if (int r2 = launch_status("cand_partition")) return r2;
    // kernel 3: per-(image, class) greedy NMS
    hipLaunchKernelGGL(nms_kernel<false>, dim3((uint32_t)nms_grid), dim3(waves * WAVE), nms_lds, st, (const float4 *)anchors,
                       rb, N, C, do_clip, hi_x, hi_y, nms_thr, max_ret, nms_keys,
                       L.cand_stride, seg_start, (const int *)coop_count, (const int *)coop_list, coop_blocks,
                       fin_count, fin_keys, L.fin_stride, (uint32_t)segs, waves, (uint32_t)per_wave, sort_cap, kept_alias,
                       (const int *)nullptr, fa, (hdr_word *)nullptr, (uint32_t)nms_grid, nms_dbg);
    if (int r3 = launch_status("nms")) return r3;
    // kernel 4: final order, top max_ret, zero padding; in pass 1 also the partition of pass 2 for the images that need it
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(FIN_THREADS), fin_part_lds, st, (const float4 *)anchors,
                       rb, N, do_clip, hi_x, hi_y, max_ret, fin_count, fin_keys,
                       L.fin_stride, (uint32_t)fin_keys_lds, (float4 *)out_boxes, out_scores, out_labels, out_keep_anchor,
                       out_count, passes == 1 ? hdr : (hdr_word *)nullptr, kpass, state,
                       (const int *)cand_count, keys_a, keys_b, L.cand_stride, C, (uint32_t)pre_nms_topk, seg_start, coop_count + 1,
                       coop_list);
    if (int r4 = launch_status("finalize")) return r4;
    if (passes == 2) {
        // pass 2: the NMS of the images whose tier was not enough, with their final top-k in the same launch
        hipLaunchKernelGGL(nms_kernel<true>, dim3(nms2_grid), dim3(waves * WAVE), nms2_lds, st, (const float4 *)anchors,
                           rb, N, C, do_clip, hi_x, hi_y, nms_thr, max_ret, nms_keys,
                           L.cand_stride, seg_start, (const int *)(coop_count + 1), (const int *)coop_list, coop_blocks,
                           fin_count, fin_keys, L.fin_stride, (uint32_t)segs, waves, (uint32_t)per_wave, sort_cap, kept_alias,
                           (const int *)state, fa, hdr, (uint32_t)nms_grid, nms_dbg);
        if (int r5 = launch_status("nms (pass 2)")) return r5;
    }
    return 0;
}

}  // namespace

extern "C" {

int odbox_decode_nms(const float *anchors, const float *regress_or_boxes, const float *scores, int B, int N,
                     int C, float img_h, float img_w, int do_clip, float score_thr, float iou_thr, int max_ret,
                     int pre_nms_topk, float *out_boxes, float *out_scores, int32_t *out_labels,
                     int32_t *out_keep_anchor, int32_t *out_count, void *workspace, size_t workspace_bytes,
                     void *stream)
{
    return decode_nms_impl("odbox_decode_nms", anchors, regress_or_boxes, ODBOX_DTYPE_F32, scores, ODBOX_DTYPE_F32, B, N, C,
                           img_h, img_w, do_clip, score_thr, iou_thr, max_ret, pre_nms_topk, out_boxes, out_scores,
                           out_labels, out_keep_anchor, out_count, workspace, workspace_bytes, stream);
}

int odbox_decode_nms_mixed(const float *anchors, const void *regress_or_boxes, int box_dtype, const void *scores,
                           int score_dtype, int B, int N, int C, float img_h, float img_w, int do_clip, float score_thr,
                           float iou_thr, int max_ret, int pre_nms_topk, float *out_boxes, float *out_scores,
                           int32_t *out_labels, int32_t *out_keep_anchor, int32_t *out_count, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    return decode_nms_impl("odbox_decode_nms_mixed", anchors, regress_or_boxes, box_dtype, scores, score_dtype, B, N, C,
                           img_h, img_w, do_clip, score_thr, iou_thr, max_ret, pre_nms_topk, out_boxes, out_scores,
                           out_labels, out_keep_anchor, out_count, workspace, workspace_bytes, stream);
}

}  // extern "C"
