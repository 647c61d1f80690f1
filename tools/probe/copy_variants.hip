// What does this HBM give the simplest kernels?  Streaming copy / read / write of a 4 GB array in several
// shapes and cache policies; prints GB/s each (bytes moved / time).  The scatter passes are priced against
// the best of these.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/copy_variants.hip -o tools/probe/copy_variants
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3 };
template <int POL> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    u32x4 v;
    if (POL == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POL == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (POL == PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// MODE 0 copy, 1 read, 2 write.  CHUNK: each block sweeps a contiguous piece (else grid-stride).  U vectors in flight.
template <int LP, int SP, int MODE, bool CHUNK, int U>
__global__ __launch_bounds__(256) void k(u32x4* __restrict__ dst, const u32x4* __restrict__ src, uint64_t nvec, uint32_t* sink) {
    uint64_t i, end, stride;
    if (CHUNK) {
        const uint64_t piece = (nvec + gridDim.x - 1) / gridDim.x;
        i = (uint64_t)blockIdx.x * piece + threadIdx.x;
        end = (uint64_t)(blockIdx.x + 1) * piece < nvec ? (uint64_t)(blockIdx.x + 1) * piece : nvec;
        stride = 256;
    } else {
        i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        end = nvec;
        stride = (uint64_t)gridDim.x * 256;
    }
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < end; i += U * stride) {
        u32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld<LP>(src + i + u * stride);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = (u32x4){(uint32_t)i, 1u, 2u, 3u};
        }
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) st<SP>(dst + i + u * stride, v[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) *sink = 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct Case { std::string name; void (*fn)(u32x4*, const u32x4*, uint64_t, uint32_t*); int mode; int blocks_per_cu; };

int main() {
    const uint64_t bytes = 4000000000ull / 16 * 16, nvec = bytes / 16;
    u32x4 *a, *b;
    uint32_t* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Case> cases = {
        {"copy plain/plain stride U4 x8", k<PLAIN, PLAIN, 0, false, 4>, 0, 8},
        {"copy plain/plain stride U4 x16", k<PLAIN, PLAIN, 0, false, 4>, 0, 16},
        {"copy plain/plain stride U4 x32", k<PLAIN, PLAIN, 0, false, 4>, 0, 32},
        {"copy plain/plain stride U8 x8", k<PLAIN, PLAIN, 0, false, 8>, 0, 8},
        {"copy plain/plain chunk U4 x8", k<PLAIN, PLAIN, 0, true, 4>, 0, 8},
        {"copy plain/plain chunk U8 x8", k<PLAIN, PLAIN, 0, true, 8>, 0, 8},
        {"copy nt/nt stride U4 x8", k<NT, NT, 0, false, 4>, 0, 8},
        {"copy nt/nt chunk U8 x8", k<NT, NT, 0, true, 8>, 0, 8},
        {"copy plain/nt stride U4 x8", k<PLAIN, NT, 0, false, 4>, 0, 8},
        {"copy nt/plain stride U4 x8", k<NT, PLAIN, 0, false, 4>, 0, 8},
        {"copy plain/sc1 stride U4 x8", k<PLAIN, SC1, 0, false, 4>, 0, 8},
        {"copy sc1/sc1 stride U4 x8", k<SC1, SC1, 0, false, 4>, 0, 8},
        {"copy plain/sc0sc1 stride U4 x8", k<PLAIN, SC0SC1, 0, false, 4>, 0, 8},
        {"copy nt/sc1 stride U4 x8", k<NT, SC1, 0, false, 4>, 0, 8},
        {"read plain stride U4 x8", k<PLAIN, PLAIN, 1, false, 4>, 1, 8},
        {"read plain stride U8 x8", k<PLAIN, PLAIN, 1, false, 8>, 1, 8},
        {"read plain chunk U8 x8", k<PLAIN, PLAIN, 1, true, 8>, 1, 8},
        {"read plain chunk U8 x4", k<PLAIN, PLAIN, 1, true, 8>, 1, 4},
        {"read nt stride U8 x8", k<NT, PLAIN, 1, false, 8>, 1, 8},
        {"read nt chunk U8 x8", k<NT, PLAIN, 1, true, 8>, 1, 8},
        {"read sc1 stride U8 x8", k<SC1, PLAIN, 1, false, 8>, 1, 8},
        {"write plain stride U4 x8", k<PLAIN, PLAIN, 2, false, 4>, 2, 8},
        {"write nt stride U4 x8", k<PLAIN, NT, 2, false, 4>, 2, 8},
        {"write sc1 stride U4 x8", k<PLAIN, SC1, 2, false, 4>, 2, 8},
        {"write nt chunk U8 x8", k<PLAIN, NT, 2, true, 8>, 2, 8},
    };
    for (auto& c : cases) {
        const dim3 grid(256 * c.blocks_per_cu);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(c.fn, grid, dim3(256), 0, 0, b, a, nvec, sink);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(c.fn, grid, dim3(256), 0, 0, b, a, nvec, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 8;
        const double moved = c.mode == 0 ? 2.0 * bytes : (double)bytes;
        printf("%-36s %7.3f ms  %7.1f GB/s\n", c.name.c_str(), ms, moved / (ms * 1e-3) / 1e9);
    }
    // the runtime's own device-to-device copy
    for (int w = 0; w < 2; ++w) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 8; ++r) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-36s %7.3f ms  %7.1f GB/s\n", "hipMemcpyAsync D2D", ms / 8, 2.0 * bytes / (ms / 8 * 1e-3) / 1e9);
    return 0;
}
