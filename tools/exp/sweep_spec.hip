// Feasibility of a sweep-ordered match+encode (configs[1] shapes): can a chip-wide lockstep sweep over the label
// tensor keep its 34.7 us (fill_shapes.hip, variant K) when OTHER waves of the same block do the matching and
// hand the store waves the few non-zero elements through LDS?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sweep_spec sweep_spec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PIECE = 1024;        // floats per piece (4 KB)
constexpr int RPP = 16;            // rows per piece at most
constexpr int VMK = 12;            // stores a store wave keeps in flight behind its progress counter
constexpr int RING = 64;           // pieces of match results a block keeps in LDS

__device__ __forceinline__ float pair_iou(const float4 a, float a_area, const float4 g, float g_area)
{
    float ix0 = fmaxf(a.x, g.x), ix1 = fminf(a.z, g.z), iy0 = fmaxf(a.y, g.y), iy1 = fminf(a.w, g.w);
    float i_area = fmaxf(ix1 - ix0, 0.0f) * fmaxf(iy1 - iy0, 0.0f);
    float u_area = ((a_area + g_area) - i_area) + 1e-8f;
    return i_area / u_area;
}

// SW store waves (each writes PIECE/SW floats of every piece), MW matcher waves
template <int SW, int MW, bool NT, int MODE>
__global__ __launch_bounds__(64 * (SW + MW)) void sweep_me(const float4 *__restrict__ anchors, const float4 *__restrict__ gt,
                                                          const int *__restrict__ gt_labels, int N, int G, int C, uint32_t rows,
                                                          float *__restrict__ out_reg, float *__restrict__ out_lab, uint32_t pieces)
{
    __shared__ uint32_t s_info[RING][RPP + 1];
    __shared__ volatile uint32_t s_ready[RING];       // piece i of this block is matched: value i + 1
    __shared__ volatile uint32_t s_done[SW];          // store wave w has consumed pieces < s_done[w]
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t d = (uint32_t)C + 1u;
    if (tid < RING) s_ready[tid] = 0u;
    if (tid < SW) s_done[tid] = 0u;
    __syncthreads();
    const uint32_t mine = (pieces - blockIdx.x + gridDim.x - 1) / gridDim.x;      // pieces of this block
    if (wave < SW) {
        // ---- store waves ----
        __builtin_amdgcn_s_setprio(3);
        constexpr uint32_t PER = PIECE / SW;           // floats per wave per piece
        for (uint32_t i = 0; i < mine; ++i) {
            const uint32_t c = blockIdx.x + i * gridDim.x;
            const uint32_t slot = i % RING;
            if (MODE == 0) while (s_ready[slot] != i + 1u) __builtin_amdgcn_s_sleep(1);
            const uint32_t f0 = c * PIECE;
            const uint32_t r0 = f0 / d;
#pragma unroll
            for (uint32_t k = 0; k < PER / 256; ++k) {
                const uint32_t f = f0 + wave * PER + k * 256u + lane * 4u;
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (MODE == 0) {
                    uint32_t r = f / d, col = f - r * d;
                    uint32_t ia = s_info[slot][r - r0], ib = s_info[slot][r - r0 + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t cc = col + e;
                        uint32_t inf = ia;
                        if (cc >= d) { cc -= d; inf = ib; }
                        const int state = (int)(inf & 3u) - 1;                  // -1 / 0 / 1
                        const uint32_t lab = inf >> 2;                         // label + 1, 0 = none
                        float x = cc == d - 1 ? (float)state : (cc + 1u == lab ? 1.0f : 0.0f);
                        v[e] = x;
                    }
                }
                if ((size_t)f + 3 < (size_t)rows * d) {
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(out_lab + f));
                    else *reinterpret_cast<v4f *>(out_lab + f) = v;
                }
            }
            if (MODE == 0 && lane == 0) s_done[wave] = i + 1u;
            if (MODE == 1) {
                // stores complete in order: all but the last VMK store instructions of this wave have reached L2
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMK) : "memory");
                constexpr uint32_t lag = (VMK + PER / 256 - 1) / (PER / 256);
                if (lane == 0 && i >= lag) s_done[wave] = i + 1u - lag;
            }
        }
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) s_done[wave] = mine;
        }
        return;
    }
    if (MODE == 2) return;
    // ---- matcher waves: batch t = pieces 4t .. 4t+3 of this block, lane = (piece, row) ----
    const uint32_t mw = wave - SW;
    const uint32_t j = lane >> 4, rr = lane & 15u;
    for (uint32_t t = mw; t * 4u < mine; t += MW) {
        const uint32_t i = t * 4u + j;
        const bool have = i < mine;
        const uint32_t c = blockIdx.x + i * gridDim.x;
        const uint32_t f0 = c * PIECE;
        const uint32_t r0 = f0 / d;
        const uint32_t R = r0 + rr;
        const bool live = have && R < rows && (uint64_t)R * d < (uint64_t)f0 + PIECE;
        uint32_t info = 1u;            // background, no label
        if (live) {
            const uint32_t b = R / (uint32_t)N, n = R - b * (uint32_t)N;
            const float4 a = anchors[n];
            const float a_area = (a.w - a.y) * (a.z - a.x);
            const float4 *gtb = gt + (size_t)b * G;
            float best_iou = -INFINITY;
            int best = 0;
            for (int g = 0; g < G; ++g) {
                const float4 gb = gtb[g];
                const float area = (gb.w - gb.y) * (gb.z - gb.x);
                float v = pair_iou(a, a_area, gb, area);
                if (fmaxf(fmaxf(gb.x, gb.y), fmaxf(gb.z, gb.w)) < 0.0f) v = -1.0f;
                if (v > best_iou) { best_iou = v; best = g; }
            }
            const bool positive = best_iou >= 0.5f;
            bool ignore = ((best_iou > 0.4f) && !positive) || (best_iou < 0.0f);
            const int state = ignore ? -1 : (positive ? 1 : 0);
            int label = -1;
            if (positive && !ignore) label = gt_labels[(size_t)b * G + best];
            if (label < 0 || label >= C) label = -1;
            info = (uint32_t)(state + 1) | ((uint32_t)(label + 1) << 2);
            if ((uint64_t)R * d >= f0 || (rr == 0 && c == 0)) {          // the row starts in this piece: its regression row is ours
                const float4 gg = gtb[best];
                const float aw = a.z - a.x, ah = a.w - a.y;
                float *p = out_reg + (size_t)R * 5;
                p[0] = (((gg.x - a.x) / aw) - 0.0f) / 0.2f;
                p[1] = (((gg.y - a.y) / ah) - 0.0f) / 0.2f;
                p[2] = (((gg.z - a.z) / aw) - 0.0f) / 0.2f;
                p[3] = (((gg.w - a.w) / ah) - 0.0f) / 0.2f;
                p[4] = (float)state;
            }
        }
        if (MODE == 1) {
            // patch the non-zero elements of this lane's row that lie in this lane's piece, once the zero stores
            // of the piece have reached L2
            if (live && info != 1u) {
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int w = 0; w < SW; ++w) ok = ok && s_done[w] > i;
                    if (!ok) __builtin_amdgcn_s_sleep(2);
                } while (!ok);
                const int state = (int)(info & 3u) - 1;
                const uint32_t lab = info >> 2;
                const uint64_t ps = (uint64_t)R * d + d - 1, pl = (uint64_t)R * d + lab - 1;
                if (ps >= f0 && ps < (uint64_t)f0 + PIECE) out_lab[ps] = (float)state;
                if (lab && pl >= f0 && pl < (uint64_t)f0 + PIECE) out_lab[pl] = 1.0f;
            }
            continue;
        }
        // ring slot free?  (piece i - RING consumed by every store wave)
        if (have && i >= RING) {
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int w = 0; w < SW; ++w) ok = ok && s_done[w] + RING > i;
                if (!ok) __builtin_amdgcn_s_sleep(2);
            } while (!ok);
        }
        if (have) s_info[i % RING][rr] = info;
        if (have && rr == 0) s_info[i % RING][RPP] = 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (have && rr == 0) s_ready[i % RING] = i + 1u;
    }
}

static int g_iter = 0;
template <typename F>
static float timed(F launch_, hipStream_t st)
{
    auto launch = [&] { launch_(); ++g_iter; };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; ++i) launch();
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    return ms * 10.0f;   // us per launch
}

int main(int argc, char **argv)
{
    const int sets = argc > 1 ? atoi(argv[1]) : 5;
    const int B = argc > 2 ? atoi(argv[2]) : 64;
    const int N = 10647, G = 20, C = 80;
    const uint32_t rows = (uint32_t)B * N;
    const size_t lab_floats = (size_t)rows * (C + 1);
    const uint32_t pieces = (uint32_t)((lab_floats + PIECE - 1) / PIECE);
    float *lab[8], *reg[8];
    for (int i = 0; i < sets; ++i) {
        CHECK(hipMalloc(&lab[i], (size_t)pieces * PIECE * 4 + 4096));
        CHECK(hipMalloc(&reg[i], (size_t)rows * 20 + 4096));
    }
    // anchors: a 3-scale grid; GT: random boxes
    std::vector<float> ha((size_t)N * 4), hg((size_t)B * G * 4);
    std::vector<int> hl((size_t)B * G);
    srand(1);
    {
        size_t k = 0;
        for (int s : {8, 16, 32}) {
            const int fw = 416 / s;
            for (int y = 0; y < fw; ++y)
                for (int x = 0; x < fw; ++x)
                    for (int a = 0; a < 3; ++a) {
                        const float w = s * (2.0f + a), h = s * (4.0f - a), cx = (x + 0.5f) * s, cy = (y + 0.5f) * s;
                        ha[k++] = cx - w / 2; ha[k++] = cy - h / 2; ha[k++] = cx + w / 2; ha[k++] = cy + h / 2;
                    }
        }
        if (k != (size_t)N * 4) { printf("anchor count %zu\n", k / 4); return 1; }
    }
    for (size_t i = 0; i < (size_t)B * G; ++i) {
        const float cx = rand() % 416, cy = rand() % 416, w = 8 + rand() % 200, h = 8 + rand() % 200;
        hg[4 * i] = cx - w / 2; hg[4 * i + 1] = cy - h / 2; hg[4 * i + 2] = cx + w / 2; hg[4 * i + 3] = cy + h / 2;
        hl[i] = rand() % C;
    }
    float4 *da, *dg;
    int *dl;
    CHECK(hipMalloc(&da, ha.size() * 4)); CHECK(hipMalloc(&dg, hg.size() * 4)); CHECK(hipMalloc(&dl, hl.size() * 4));
    CHECK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    printf("label bytes %zu, pieces %u, %d set(s) in rotation\n", lab_floats * 4, pieces, sets);
#define L (lab[g_iter % sets])
#define R (reg[g_iter % sets])
#define RUN(SW, MW, NT, MODE, grid) \
    printf("SW %d MW %d nt %d mode %d grid %4u: %.2f us\n", SW, MW, NT, MODE, grid, \
           timed([&] { sweep_me<SW, MW, NT, MODE><<<grid, 64 * (SW + MW), 0, st>>>(da, dg, dl, N, G, C, rows, R, L, pieces); }, st)); fflush(stdout)
    for (uint32_t grid : {256u, 512u}) {
        RUN(4, 0, true, 2, grid);      // bare sweep
        RUN(4, 0, false, 2, grid);
        RUN(4, 4, true, 1, grid);
        RUN(4, 8, true, 1, grid);
        RUN(4, 12, true, 1, grid);
        RUN(4, 8, false, 1, grid);
        RUN(2, 8, true, 1, grid);
        RUN(1, 8, true, 1, grid);
        RUN(4, 4, true, 0, grid);
    }
    // check one launch against a host restatement of the label tensor's non-zeros (count only)
    {
        CHECK(hipMemsetAsync(lab[0], 0xff, lab_floats * 4, st));
        sweep_me<4, 8, true, 1><<<256, 64 * 12, 0, st>>>(da, dg, dl, N, G, C, rows, reg[0], lab[0], pieces);
        CHECK(hipStreamSynchronize(st));
        std::vector<float> h(lab_floats);
        CHECK(hipMemcpy(h.data(), lab[0], lab_floats * 4, hipMemcpyDeviceToHost));
        size_t ones = 0, neg = 0, bad = 0, state1 = 0;
        for (size_t r = 0; r < rows; ++r) {
            const float *p = &h[r * (C + 1)];
            int n1 = 0;
            for (int c = 0; c < C; ++c) { if (p[c] == 1.0f) ++n1; else if (p[c] != 0.0f) ++bad; }
            ones += n1;
            if (p[C] == 1.0f) { ++state1; if (n1 != 1) ++bad; }
            else if (p[C] == -1.0f) { ++neg; if (n1) ++bad; }
            else if (p[C] != 0.0f || n1) ++bad;
        }
        printf("check: %zu one-hot ones, %zu foreground, %zu ignored, %zu bad elements\n", ones, state1, neg, bad);
    }
    return 0;
}
