// Does gfx950 serve ds_read_b128 at a 2-byte-aligned LDS address, and at what cost?  (round 4: a time-on-K bf16 weight
// gradient would read its x fragments at a per-lane tap shift, i.e. 2-byte aligned.)
//   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_bench.hip -o tools/_build/lds_unaligned_bench && tools/_build/lds_unaligned_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__global__ void check_kernel(unsigned *out, int shift_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned addr = base + threadIdx.x * 32 + shift_bytes;
    u32x4 v = lds_read128(addr);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

template <int MODE>   // 0: aligned, 1: per-lane (lane & 7) * 2 byte shift, 2: +2 bytes for all, 3: +4 bytes for all
__global__ void time_kernel(unsigned *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)lds;
    const int lane = threadIdx.x & 63;
    unsigned sh = MODE == 1 ? (lane & 7) * 2 : MODE == 2 ? 2 : MODE == 3 ? 4 : 0;
    unsigned addr = base + (threadIdx.x & 255) * 16 + sh;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        u32x4 a, b, c, d;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                     "ds_read_b128 %3, %4 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
        acc += a + b + c + d;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    unsigned *d;
    hipMalloc(&d, 1 << 22);
    for (int sh : {0, 2, 4, 6, 10, 14}) {
        check_kernel<<<1, 64>>>(d, sh);
        std::vector<unsigned> h(256);
        hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t)
            for (int j = 0; j < 16; ++j) {
                unsigned char got = (h[t * 4 + j / 4] >> (8 * (j % 4))) & 255, want = (unsigned char)((t * 32 + sh + j) * 7 + 3);
                bad += got != want;
            }
        printf("ds_read_b128 at +%d bytes: %s (%d wrong bytes)%s\n", sh, bad ? "WRONG" : "correct", bad,
               hipGetLastError() == hipSuccess ? "" : " [launch error]");
    }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 512;
    auto run = [&](auto kern, const char *name) {
        kern<<<blocks, 512>>>(d, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        kern<<<blocks, 512>>>(d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double bytes = (double)blocks * 512 * iters * 4 * 16;
        printf("%-34s %8.3f ms  %7.1f TB/s LDS read (chip)\n", name, ms, bytes / ms / 1e9);
    };
    run(time_kernel<0>, "aligned b128");
    run(time_kernel<3>, "+4 bytes (dword aligned) b128");
    run(time_kernel<2>, "+2 bytes b128");
    run(time_kernel<1>, "per-lane (lane&7)*2 bytes b128");
    return 0;
}
