// How fast can 1.13 GB be READ (config 5's score tensor)?  Plain 16-byte loads, with and without nt,
// persistent grids of several sizes.  Tuning aid for score_scan_kernel (build like fill_shapes.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void read_grid_stride(const v4f *p, size_t nvec, float *out)
{
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        v4f x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const v4f *src = p + v + u * stride;
            if (MODE == 0) x[u] = *src;
            else x[u] = __builtin_nontemporal_load(src);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += fmaxf(fmaxf(x[u].x, x[u].y), fmaxf(x[u].z, x[u].w));
    }
    for (; v < nvec; v += stride) { const v4f x = p[v]; acc += x.x; }
    if (acc == 12345.678f) out[0] = acc;     // keep the loads alive
}

template <typename F>
static float timed(F launch, hipStream_t st)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 50.0f;   // us per launch
}

int main(int argc, char **argv)
{
    // default: config 5's score tensor (1.131 GB); argv[1] = size in MB (240 = what the loss forward reads)
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) * 1000000ull / 16 * 16 : 64ull * 49104 * 90 * 4;
    const size_t nvec = bytes / 16;
    v4f *p;
    float *out;
    CHECK(hipMalloc(&p, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(p, 0, bytes));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    for (uint32_t g : {1024u, 2048u, 4096u, 8192u, 16384u}) {
        printf("plain u4 %6u blocks  %.1f us\n", g, timed([&] { read_grid_stride<0, 4><<<g, 256, 0, st>>>(p, nvec, out); }, st));
        printf("nt    u4 %6u blocks  %.1f us\n", g, timed([&] { read_grid_stride<1, 4><<<g, 256, 0, st>>>(p, nvec, out); }, st));
    }
    printf("plain u8   2048 blocks  %.1f us\n", timed([&] { read_grid_stride<0, 8><<<2048, 256, 0, st>>>(p, nvec, out); }, st));
    printf("nt    u8   2048 blocks  %.1f us\n", timed([&] { read_grid_stride<1, 8><<<2048, 256, 0, st>>>(p, nvec, out); }, st));
    printf("nt    u2   4096 blocks  %.1f us\n", timed([&] { read_grid_stride<1, 2><<<4096, 256, 0, st>>>(p, nvec, out); }, st));
    return 0;
}
