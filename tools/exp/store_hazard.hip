// Why the library uses compiler-visible stores only.  Compile to assembly and compare the two kernels:
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/exp/store_hazard.hip -o - | grep -A3 global_store_dwordx4
// After its OWN 128-bit store hipcc puts `s_nop 1` in front of the VALU instruction that overwrites the store data
// registers (gfx950 hazard: a VMEM store of more than 64 bits reads its data a few cycles after issue); after an
// inline-asm store it inserts nothing, and the next v_pk_mul_f32 overwrites v[4:5] while the store may still read them.
#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_asm(v4f *dst, const float *src, float s)
{
    v4f v = {src[threadIdx.x], src[threadIdx.x + 64], src[threadIdx.x + 128], src[threadIdx.x + 192]};
    v4f *p = dst + threadIdx.x;
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    v = v * s;
    asm volatile("global_store_dwordx4 %0, %1, off offset:1024 sc1 nt" ::"v"(p), "v"(v) : "memory");
    v = v * s;
    asm volatile("global_store_dwordx4 %0, %1, off offset:2048 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
__global__ void k_cpp(v4f *dst, const float *src, float s)
{
    v4f v = {src[threadIdx.x], src[threadIdx.x + 64], src[threadIdx.x + 128], src[threadIdx.x + 192]};
    v4f *p = dst + threadIdx.x;
    __builtin_nontemporal_store(v, p);
    v = v * s;
    __builtin_nontemporal_store(v, p + 64);
    v = v * s;
    __builtin_nontemporal_store(v, p + 128);
}
