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

static int g_iter = 0;      // launch counter: the experiments write buffer g_iter % sets
// E: tile chunks with the number of resident blocks per CU limited by dynamic LDS (window of tiles being written)
__global__ __launch_bounds__(256) void fill_block_chunk_occ(v4f *p, uint32_t per_block, size_t nvec)
{
    extern __shared__ float dyn[];
    if (per_block == 0xffffffffu) dyn[threadIdx.x] = 0.f;
    const size_t base = (size_t)blockIdx.x * per_block;
    const v4f z = {0, 0, 0, 0};
    for (uint32_t v = threadIdx.x; v < per_block; v += 256)
        if (base + v < nvec) p[base + v] = z;
}
// F: persistent blocks striding over tiles: block k writes tiles k, k + grid, ...
template <bool NT>
__global__ __launch_bounds__(256) void fill_tiles_persistent(v4f *p, uint32_t per_block, uint32_t tiles, size_t nvec)
{
    const v4f z = {0, 0, 0, 0};
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = (size_t)t * per_block;
        for (uint32_t v = threadIdx.x; v < per_block; v += 256)
            if (base + v < nvec) {
                if (NT) __builtin_nontemporal_store(z, p + base + v);
                else p[base + v] = z;
            }
    }
}

// G: grid-stride sweep whose blocks lose lockstep: MODE 0 = every `period` pieces the block burns `burn` clocks
// (a stand-in for the match of one tile), MODE 1 = the same with a per-(block, step) pseudo-random length,
// MODE 2 = no burn, but XCD x owns a contiguous eighth of every grid-wide row instead of every 8th piece
template <int MODE>
__global__ __launch_bounds__(256) void fill_sweep_jitter(v4f *p, size_t nvec, uint32_t period, uint32_t burn)
{
    const v4f z = {0, 0, 0, 0};
    uint32_t it = 0;
    uint32_t blk = blockIdx.x;
    if (MODE == 2) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (size_t v = (size_t)blk * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256, ++it) {
        p[v] = z;
        if (MODE != 2 && it % period == period - 1) {
            uint32_t n = burn;
            if (MODE == 1) n = (burn * (((blockIdx.x * 2654435761u + it * 40503u) >> 16) & 255u)) >> 7;   // 0 .. 2 x burn
            const long long t0 = clock64();
            while (clock64() - t0 < (long long)n) __builtin_amdgcn_s_sleep(1);
        }
    }
}

// H: bigger blocks on proportionally bigger chunks (NT: nontemporal stores): BS threads write BS/256 tiles
template <int BS, bool NT>
__global__ __launch_bounds__(BS) void fill_big_block(v4f *p, uint32_t per_block, size_t nvec)
{
    const size_t base = (size_t)blockIdx.x * per_block;
    const v4f z = {0, 0, 0, 0};
    for (uint32_t v = threadIdx.x; v < per_block; v += BS)
        if (base + v < nvec) {
            if (NT) __builtin_nontemporal_store(z, p + base + v);
            else p[base + v] = z;
        }
}


// K: persistent blocks in lockstep over small block-contiguous chunks: block k writes chunks k, k + grid, ...
// (chunk_vec vectors each; `burn` clock64 ticks of VALU-free waiting per chunk stand in for a match)
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunks_persistent(v4f *p, uint32_t chunk_vec, uint32_t chunks, size_t nvec, uint32_t burn)
{
    const v4f z = {0, 0, 0, 0};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        if (burn) {
            const long long t0 = clock64();
            while (clock64() - t0 < (long long)burn) __builtin_amdgcn_s_sleep(1);
        }
        const size_t base = (size_t)c * chunk_vec;
        for (uint32_t v = threadIdx.x; v < chunk_vec; v += 256)
            if (base + v < nvec) {
                if (NT) __builtin_nontemporal_store(z, p + base + v);
                else p[base + v] = z;
            }
    }
}

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
    return ms * 10.0f;   // us per launch
}

// usage: fill_shapes [sets [images]]   sets > 1: consecutive launches write different buffers (sets x 221 MB in
// rotation), so nothing is absorbed by the 256 MB Infinity Cache
int main(int argc, char **argv)
{
    const int sets = argc > 1 ? atoi(argv[1]) : 1;
    const size_t images = argc > 2 ? (size_t)atoi(argv[2]) : 64;
    const size_t bytes = images * 10647 * 81 * 4;           // config 2's label tensor: 220.8 MB at 64 images
    const size_t nvec = bytes / 16;
    v4f *bufs[16];
    for (int i = 0; i < sets && i < 16; ++i) CHECK(hipMalloc(&bufs[i], bytes + 4096));
#define p (bufs[g_iter % sets])
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const uint32_t per_block = 256 * 81 / 4;                // one 256-anchor tile: 5184 vectors = 83 KB
    const uint32_t blocks = (uint32_t)((nvec + per_block - 1) / per_block);
    printf("bytes %zu, tile blocks %u, %d buffer(s) in rotation\n", bytes, blocks, sets);
    for (uint32_t g : {256u, 512u, 1024u})
        for (uint32_t cv : {256u, 512u, 1024u, 1296u, 2592u}) {          // 4 KB, 8 KB, 16 KB, one / two 64-anchor tiles
            const uint32_t chunks = (uint32_t)((nvec + cv - 1) / cv);
            printf("K persistent %4u blocks x %5u B chunks: %.2f us  nt %.2f us  nt+burn50 %.2f us  nt+burn100 %.2f us\n", g, cv * 16,
                   timed([&] { fill_chunks_persistent<false><<<g, 256, 0, st>>>(p, cv, chunks, nvec, 0); }, st),
                   timed([&] { fill_chunks_persistent<true><<<g, 256, 0, st>>>(p, cv, chunks, nvec, 0); }, st),
                   timed([&] { fill_chunks_persistent<true><<<g, 256, 0, st>>>(p, cv, chunks, nvec, 50); }, st),
                   timed([&] { fill_chunks_persistent<true><<<g, 256, 0, st>>>(p, cv, chunks, nvec, 100); }, st));
        }
    if (argc > 3) return 0;
    printf("hipMemsetAsync            %.2f us\n", timed([&] { CHECK(hipMemsetAsync(p, 0, bytes, st)); }, st));
    printf("A block chunk (tile)      %.2f us\n", timed([&] { fill_block_chunk<<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    for (uint32_t pb : {1024u, 2048u, 10368u, 20736u}) {
        const uint32_t nb = (uint32_t)((nvec + pb - 1) / pb);
        printf("A block chunk %6u vec  %.2f us (%u blocks)\n", pb, timed([&] { fill_block_chunk<<<nb, 256, 0, st>>>(p, pb, nvec); }, st), nb);
    }
    for (uint32_t g : {256u, 512u, 1024u, 2048u, 4096u, 8192u})
        printf("B grid stride %5u blocks %.2f us\n", g, timed([&] { fill_grid_stride<<<g, 256, 0, st>>>(p, nvec); }, st));
    for (uint32_t lds : {20u, 36u, 50u, 70u, 100u})
        printf("E tile chunk, %3u KB LDS/block %.2f us\n", lds, timed([&] { fill_block_chunk_occ<<<blocks, 256, lds * 1024, st>>>(p, per_block, nvec); }, st));
    for (uint32_t g : {256u, 512u, 768u, 1024u, 2048u}) {
        printf("F persistent tiles %4u blocks %.2f us", g, timed([&] { fill_tiles_persistent<false><<<g, 256, 0, st>>>(p, per_block, blocks, nvec); }, st));
        printf("   nt %.2f us\n", timed([&] { fill_tiles_persistent<true><<<g, 256, 0, st>>>(p, per_block, blocks, nvec); }, st));
    }
    for (uint32_t g : {256u, 512u}) {
        printf("G sweep %u blocks: XCD-contiguous rows %.2f us", g, timed([&] { fill_sweep_jitter<2><<<g, 256, 0, st>>>(p, nvec, 20, 0); }, st));
        for (uint32_t burn : {1000u, 3000u, 6000u})      // clock64 ticks at 100 MHz: 10 / 30 / 60 us?  (s_memtime domain) -- printed relative
            printf("   burn %u fixed %.2f us, random %.2f us", burn, timed([&] { fill_sweep_jitter<0><<<g, 256, 0, st>>>(p, nvec, 20, burn); }, st),
                   timed([&] { fill_sweep_jitter<1><<<g, 256, 0, st>>>(p, nvec, 20, burn); }, st));
        printf("\n");
    }
    {
        const uint32_t pb2 = per_block * 2, pb4 = per_block * 4, pbh = per_block / 2, pbq = per_block / 4;
        printf("H 64 thr x 1/4 tile %.2f us  nt %.2f us\n", timed([&] { fill_big_block<64, false><<<(uint32_t)((nvec + pbq - 1) / pbq), 64, 0, st>>>(p, pbq, nvec); }, st),
               timed([&] { fill_big_block<64, true><<<(uint32_t)((nvec + pbq - 1) / pbq), 64, 0, st>>>(p, pbq, nvec); }, st));
        printf("H 128 thr x 1/2 tile %.2f us  nt %.2f us\n", timed([&] { fill_big_block<128, false><<<(uint32_t)((nvec + pbh - 1) / pbh), 128, 0, st>>>(p, pbh, nvec); }, st),
               timed([&] { fill_big_block<128, true><<<(uint32_t)((nvec + pbh - 1) / pbh), 128, 0, st>>>(p, pbh, nvec); }, st));
        printf("H 512 thr x 2 tiles %.2f us  nt %.2f us\n", timed([&] { fill_big_block<512, false><<<(uint32_t)((nvec + pb2 - 1) / pb2), 512, 0, st>>>(p, pb2, nvec); }, st),
               timed([&] { fill_big_block<512, true><<<(uint32_t)((nvec + pb2 - 1) / pb2), 512, 0, st>>>(p, pb2, nvec); }, st));
        printf("H 1024 thr x 4 tiles %.2f us  nt %.2f us\n", timed([&] { fill_big_block<1024, false><<<(uint32_t)((nvec + pb4 - 1) / pb4), 1024, 0, st>>>(p, pb4, nvec); }, st),
               timed([&] { fill_big_block<1024, true><<<(uint32_t)((nvec + pb4 - 1) / pb4), 1024, 0, st>>>(p, pb4, nvec); }, st));
    }
    printf("C 64 B per thread (tile)  %.2f us\n", timed([&] { fill_thread64<<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc1                     %.2f us\n", timed([&] { fill_block_chunk_asm<0><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D nt                      %.2f us\n", timed([&] { fill_block_chunk_asm<1><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc0 sc1                 %.2f us\n", timed([&] { fill_block_chunk_asm<2><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    printf("D sc0                     %.2f us\n", timed([&] { fill_block_chunk_asm<3><<<blocks, 256, 0, st>>>(p, per_block, nvec); }, st));
    return 0;
}
