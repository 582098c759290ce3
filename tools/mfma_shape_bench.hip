// mfma_shape_bench.hip — guide rule 28 ("build both bf16 MFMA shapes at the same wave tile, keep the faster by wall
// clock") as a standalone measurement: the inner loop of csrc/conv1d_bf16_ring.hip (wave tile 160 time steps x 64
// channels, fragments read from LDS with ds_read_b128, two waves per SIMD, one workgroup of eight waves per CU) once with
// v_mfma_f32_32x32x16_bf16 (what the product kernels use) and once with v_mfma_f32_16x16x32_bf16, nothing else in the
// loop: no global traffic, no ring, no barriers.  Both variants read the SAME number of LDS bytes per flop (7 fragment
// reads per 10 MFMAs of 32x32x16 = 14 per 40 of 16x16x32) and hold the same 160 accumulator registers; what differs is
// the number of matrix instructions issued (2x for 16x16x32) and the K depth per instruction.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o gpurun_out/mfma_shape_bench
//   ./gpurun_out/mfma_shape_bench            (prints one JSON line per variant)
//
// Diagnostic only: not part of libecg_hip.so, nothing in the product path calls it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NW = 8, NT = 64 * NW;
constexpr int ROWS = 4 * 160 + 64;          // x image rows (32 B each): 4 time slices of 160 + tap halo
constexpr int XBYTES = ROWS * 32;
constexpr int WBYTES = 16 * 128 * 32;       // 16 tap slices of [128 channels][16] bf16
constexpr int TAPS = 16;

#define CHECK(e) do { hipError_t _r = (e); if (_r != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_r)); exit(1); } } while (0)

__device__ __forceinline__ void fill_lds(unsigned char *lds, const unsigned *src) {
    unsigned *l = reinterpret_cast<unsigned *>(lds);
    for (int i = threadIdx.x; i < (XBYTES + WBYTES) / 4; i += NT) l[i] = src[i];
    __syncthreads();
}

// 32x32x16: A = x fragment (rows = time, lane l31 -> row, half -> 8 of the 16 channels), B = weight fragment
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void loop_32x32x16(const unsigned *src, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES + WBYTES];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int wt = (wave & 3) * 160, wco = (wave >> 2) * 64;
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int xo = (wt + l31) * 32 + half * 16, wo = XBYTES + (wco + l31) * 32 + half * 16;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(xo), "+v"(wo));           // opaque per trip: the fragment reads stay IN the loop
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(&lds[wo + k * 4096]);
            bf16x8 w1 = *reinterpret_cast<const bf16x8 *>(&lds[wo + k * 4096 + 1024]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                bf16x8 xa = *reinterpret_cast<const bf16x8 *>(&lds[xo + k * 32 + j * 1024]);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, w0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, w1, acc[1][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * NT + threadIdx.x] = s;
}

// 16x16x32: lane l15 -> row, kq = lane >> 4 -> 8 of the 32 K elements = (tap pair member kq >> 1, channel half kq & 1):
// one instruction covers TWO taps of the 16-channel chunk; 10 row tiles x 4 column tiles per wave
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void loop_16x16x32(const unsigned *src, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES + WBYTES];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, l15 = lane & 15;
    const int wt = (wave & 3) * 160, wco = (wave >> 2) * 64;
    f32x4 acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    int xo = (wt + l15 + (kq >> 1)) * 32 + (kq & 1) * 16;                       // second tap of the pair = one row further
    int wo = XBYTES + (kq >> 1) * 4096 + (wco + l15) * 32 + (kq & 1) * 16;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(xo), "+v"(wo));
#pragma unroll
        for (int k = 0; k < TAPS; k += 2) {
            bf16x8 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const bf16x8 *>(&lds[wo + k * 4096 + i * 512]);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                bf16x8 xa = *reinterpret_cast<const bf16x8 *>(&lds[xo + k * 32 + j * 512]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, w[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * NT + threadIdx.x] = s;
}

// ---- fp32 MFMA (the parity path: csrc/conv1d_mfma.hip): v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 ----
// wave tile 64 x 64 (four 32x32 or sixteen 16x16 accumulators = 64 registers), four waves per workgroup, two workgroups
// per CU (two waves per SIMD), operands read from LDS one float per lane and k step, as the product kernels do
constexpr int FK = 64;                       // k steps held in LDS: A [FK][128] + B [FK][128] floats = 64 KB
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void loop_f32_32x32x2(const unsigned *src, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * FK * 128];
    for (int i = threadIdx.x; i < 2 * FK * 128; i += 256) lds[i] = __uint_as_float((src[i % ((XBYTES + WBYTES) / 4)] & 0x807FFFFFu) | 0x3F000000u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int ao = half * 128 + wm + l31, bo = FK * 128 + half * 128 + wn + l31;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(ao), "+v"(bo));
#pragma unroll
        for (int k = 0; k < FK; k += 2) {
            const float a0 = lds[ao + k * 128], a1 = lds[ao + k * 128 + 32];
            const float b0 = lds[bo + k * 128], b1 = lds[bo + k * 128 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void loop_f32_16x16x4(const unsigned *src, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * FK * 128];
    for (int i = threadIdx.x; i < 2 * FK * 128; i += 256) lds[i] = __uint_as_float((src[i % ((XBYTES + WBYTES) / 4)] & 0x807FFFFFu) | 0x3F000000u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, l15 = lane & 15;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    int ao = kq * 128 + wm + l15, bo = FK * 128 + kq * 128 + wn + l15;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(ao), "+v"(bo));
#pragma unroll
        for (int k = 0; k < FK; k += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = lds[ao + k * 128 + 16 * i]; b[i] = lds[bo + k * 128 + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static void run_f32(const unsigned *src, float *out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int wgs = 512, reps = 10;
    const double flops = 2.0 * 64 * 64 * FK * (double)iters * 4 * wgs;
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(loop_f32_32x32x2, dim3(wgs), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(loop_f32_16x16x4, dim3(wgs), dim3(256), 0, 0, src, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double avg = sum / reps;
        printf("{\"loop\": \"%s\", \"wave_tile\": \"64x64\", \"waves_per_simd\": 2, \"iters\": %d, \"avg_ms\": %.4f, \"best_ms\": %.4f, "
               "\"tflops_avg\": %.1f, \"tflops_best\": %.1f, \"frac_of_157.3\": %.3f}\n",
               variant == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", iters, avg, best,
               flops / avg / 1e9, flops / best / 1e9, flops / avg / 1e9 / 157.3);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400, wgs = 256, reps = 10;
    std::vector<unsigned> h((XBYTES + WBYTES) / 4);
    unsigned st = 12345u;
    for (auto &v : h) {                     // random bf16 pairs in roughly [-2, 2]: realistic toggling for the power / clock behaviour
        st = st * 1664525u + 1013904223u;
        const unsigned a = 0x3F00u | ((st >> 9) & 0x80FFu), b = 0x3F00u | ((st >> 17) & 0x80FFu);
        v = a | (b << 16);
    }
    unsigned *src; float *out;
    CHECK(hipMalloc(&src, h.size() * 4));
    CHECK(hipMalloc(&out, (size_t)wgs * NT * 4));
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double flops = 2.0 * 160 * 64 * 16 * TAPS * (double)iters * NW * wgs;     // per launch
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(loop_32x32x16, dim3(wgs), dim3(NT), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(loop_16x16x32, dim3(wgs), dim3(NT), 0, 0, src, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double avg = sum / reps;
        printf("{\"loop\": \"%s\", \"wave_tile\": \"160x64\", \"waves_per_simd\": 2, \"iters\": %d, \"avg_ms\": %.4f, \"best_ms\": %.4f, "
               "\"tflops_avg\": %.1f, \"tflops_best\": %.1f, \"frac_of_2500\": %.3f}\n",
               variant == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", iters, avg, best,
               flops / avg / 1e9, flops / best / 1e9, flops / avg / 1e9 / 2500.0);
    }
    run_f32(src, out, iters);
    return 0;
}
