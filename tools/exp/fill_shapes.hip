// How fast can 245 MB of zeros be written, and with which access shape?  (Tuning aid for the zero
// stream of match_encode_kernel; build: hipcc --offload-arch=gfx950 -O3 -o fill_shapes fill_shapes.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: block-contiguous chunk of `per_block` vectors, threads interleaved (what match_encode does)
__global__ __launch_bounds__(256) void fill_block_chunk(v4f *p, uint32_t per_block, size_t nvec)
{
    const size_t base = (size_t)blockIdx.x * per_block;
    const v4f z = {0, 0, 0, 0};
    for (uint32_t v = threadIdx.x; v < per_block; v += 256)
        if (base + v < nvec) p[base + v] = z;
}
// B: grid-stride, persistent
__global__ __launch_bounds__(256) void fill_grid_stride(v4f *p, size_t nvec)
{
    const v4f z = {0, 0, 0, 0};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) p[v] = z;
}
// C: each thread writes 4 consecutive vectors (64 B), block-contiguous
__global__ __launch_bounds__(256) void fill_thread64(v4f *p, uint32_t per_block, size_t nvec)
{
    const size_t base = (size_t)blockIdx.x * per_block;
    const v4f z = {0, 0, 0, 0};
    for (uint32_t v = threadIdx.x * 4; v < per_block; v += 1024)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + v + k < nvec && v + k < per_block) p[base + v + k] = z;
}
// D: like A with sc1 / nt / sc0 sc1 stores
template <int MODE>
__global__ __launch_bounds__(256) void fill_block_chunk_asm(v4f *p, uint32_t per_block, size_t nvec)
{
    const size_t base = (size_t)blockIdx.x * per_block;
    const v4f z = {0, 0, 0, 0};
    for (uint32_t v = threadIdx.x; v < per_block; v += 256)
        if (base + v < nvec) {
            v4f *d = p + base + v;
            if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(z) : "memory");
            if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(d), "v"(z) : "memory");
            if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(z) : "memory");
            if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(d), "v"(z) : "memory");
        }
}

template <typename F>
static float timed(F launch, hipStream_t st)
{
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
    return ms * 10.0f;   // us per launch
}

int main()
{
    const size_t bytes = 64ull * 10647 * 81 * 4;            // config 2's label tensor: 220.8 MB
    const size_t nvec = bytes / 16;
    v4f *p;
    CHECK(hipMalloc(&p, bytes + 4096));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const uint32_t per_block = 256 * 81 / 4;                // one 256-anchor tile: 5184 vectors = 83 KB
    const uint32_t blocks = (uint32_t)((nvec + per_block - 1) / per_block);
    printf("bytes %zu, tile blocks %u\n", bytes, blocks);
    printf("hipMemsetAsync            %.2f us\n", timed([&] { CHECK(hipMemsetAsync(p, 0, bytes, st)); }, st));
    printf("A block chunk (tile)      %.2f us\n", timed([&] { fill_block_chunk<<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    for (uint32_t pb : {1024u, 2048u, 10368u, 20736u}) {
        const uint32_t nb = (uint32_t)((nvec + pb - 1) / pb);
        printf("A block chunk %6u vec  %.2f us (%u blocks)\n", pb, timed([&] { fill_block_chunk<<<nb, 256, 0, st>>>(p, pb, nvec); }, st), nb);
    }
    for (uint32_t g : {256u, 512u, 1024u, 2048u, 4096u, 8192u})
        printf("B grid stride %5u blocks %.2f us\n", g, timed([&] { fill_grid_stride<<<g, 256, 0, st>>>(p, nvec); }, st));
    printf("C 64 B per thread (tile)  %.2f us\n", timed([&] { fill_thread64<<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc1                     %.2f us\n", timed([&] { fill_block_chunk_asm<0><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D nt                      %.2f us\n", timed([&] { fill_block_chunk_asm<1><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc0 sc1                 %.2f us\n", timed([&] { fill_block_chunk_asm<2><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc0                     %.2f us\n", timed([&] { fill_block_chunk_asm<3><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    return 0;
}
