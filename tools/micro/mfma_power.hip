// Micro-benchmark (GPU box): what the chip SUSTAINS in f16 MFMA on random operands -- it runs the FRCNN step at its package power
// limit (~1300 W), so a kernel's ceiling is set by its energy per flop, not by the 2.5 PFLOP/s issue peak.
//   variant            per 16 MFMAs (16x16x32) and wave
//   mfma16 / mfma32    registers only: the two gfx950 f16 shapes (mfma16 also with its accumulators in AGPRs)
//   lds R              + R ds_read_b128 (fragment reads: the panel 3x3 kernel has 5.8, conv_gemm4 4.0)
//   dma R D            + D global_load_lds_dwordx4 of 1 KB from a 2 MiB buffer (L1 misses, L2 hits) (panel 1.1, gemm4 2.0)
// 256 workgroups x 8 waves (2 per SIMD).   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// SHAPE 16 | 32; R: ds_read_b128 per 16 MFMAs; D: LDS-DMA pieces per 16 MFMAs in halves (D2 = 2 * D: 1 -> one piece every other iteration)
template <int SHAPE, int R, int D2>
__global__ __launch_bounds__(512) void k(const _Float16 *src, const char *l2buf, unsigned l2mask, float *dst, int iters, unsigned long *clk) {
    __shared__ __attribute__((aligned(16))) char lds[65536 + 16384];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<half8 *>(lds)[i] = *reinterpret_cast<const half8 *>(src + (i * 8) % 65536);
    __syncthreads();
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const half8 *>(src + ((tid * 4 + i) * 8) % 65536);
        b[i] = *reinterpret_cast<const half8 *>(src + ((tid * 4 + i) * 8 + 32768) % 65536);
    }
    const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)lds;
    unsigned laddr = lbase + (tid & 63) * 16 + wave * 4096;
    unsigned goff = (blockIdx.x * 8 + wave) * 8192 + (tid & 63) * 16;
    unsigned long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        floatx4 acc[4][4] = {};
        half8 a2[4], b2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a2[i] = b[i], b2[i] = a[i];
        // software-pipelined like the real kernels: the NEXT iteration's fragment reads and DMA pieces are issued before this
        // iteration's 16 MFMAs and waited for after them, so nothing but power (or the LDS / vector-memory pipes) holds the MFMAs back
        auto phase = [&](half8(&ca)[4], half8(&cb)[4], half8(&na)[4], half8(&nb)[4], int it) {
            if constexpr (D2 > 0) {
                if (D2 >= 2 || (it & 1)) {
#pragma unroll
                    for (int d = 0; d < (D2 >= 2 ? D2 / 2 : 1); ++d) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(l2buf + ((goff + d * 1024) & l2mask)),
                                                         (__attribute__((address_space(3))) void *)(lds + 65536 + wave * 2048 + d * 1024), 16, 0, 0);
                    }
                    goff += 73728;                       // walks the whole buffer: every piece misses the 32 KB L1 and hits L2
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < 4)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(na[r]) : "v"(laddr), "n"(r * 1024 % 4096));
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nb[r - 4]) : "v"(laddr), "n"(r * 1024 % 4096));
            }
            if constexpr (R > 0) laddr = lbase + ((laddr - lbase + 16384) & 65535);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca[i], cb[j], acc[i][j], 0, 0, 0);
            if constexpr (R > 0)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na[0]), "+v"(na[1]), "+v"(na[2]), "+v"(na[3]), "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]));
            if constexpr (D2 > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        };
        for (int it = 0; it < iters; it += 2) {
            phase(a, b, a2, b2, it);
            phase(a2, b2, a, b, it + 1);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else if constexpr (SHAPE == 17) {
        floatx4 acc[4][4] = {};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        floatx16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    dst[blockIdx.x * blockDim.x + tid] = s + (float)lds[65536 + tid];
    if (tid == 0) {
        if (blockIdx.x == 0) {
            clk[0] = t1 - t0;
            clk[1] = r1 - r0;
        }
        clk[2 + 2 * blockIdx.x] = r0;
        clk[3 + 2 * blockIdx.x] = r1;
    }
}

typedef void (*kern_t)(const _Float16 *, const char *, unsigned, float *, int, unsigned long *);
struct Var { const char *name; kern_t f; };

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000, wgs = 256;
    std::vector<_Float16> h(65536);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    const unsigned l2bytes = 2u << 20;                       // 2 MiB: resident in every XCD's 4 MiB L2
    std::vector<_Float16> hb(l2bytes / 2);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = h[i & 65535];
    _Float16 *src;
    char *l2buf;
    float *dst;
    unsigned long *clk, hclk[2 + 512];
    CK(hipMalloc(&src, 65536 * 2));
    CK(hipMalloc(&l2buf, l2bytes));
    CK(hipMalloc(&dst, wgs * 512 * 4));
    CK(hipMalloc(&clk, sizeof(hclk)));
    CK(hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(l2buf, hb.data(), l2bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const Var vars[] = {{"mfma16 (registers only)", k<16, 0, 0>}, {"mfma16, accumulators in AGPRs", k<17, 0, 0>}, {"mfma32 (registers only)", k<32, 0, 0>},
                        {"mfma16 + 4 ds_read (gemm4 ratio)", k<16, 4, 0>}, {"mfma16 + 6 ds_read (panel ratio)", k<16, 6, 0>},
                        {"mfma16 + 8 ds_read", k<16, 8, 0>},
                        {"mfma16 + 6 ds_read + 1 DMA KB (panel)", k<16, 6, 2>}, {"mfma16 + 4 ds_read + 2 DMA KB (gemm4)", k<16, 4, 4>},
                        {"mfma16 + 4 ds_read + 0.5 DMA KB", k<16, 4, 1>}};
    for (int rep = 0; rep < 2; ++rep)
        for (const Var &v : vars) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.f, dim3(wgs), dim3(512), 0, 0, src, l2buf, l2bytes - 1, dst, iters, clk);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hclk, clk, sizeof(hclk), hipMemcpyDeviceToHost));
            unsigned long lo = ~0ul, hi = 0, dmin = ~0ul, dmax = 0;
            for (int w = 0; w < wgs; ++w) {
                lo = hclk[2 + 2 * w] < lo ? hclk[2 + 2 * w] : lo;
                hi = hclk[3 + 2 * w] > hi ? hclk[3 + 2 * w] : hi;
                const unsigned long d = hclk[3 + 2 * w] - hclk[2 + 2 * w];
                dmin = d < dmin ? d : dmin;
                dmax = d > dmax ? d : dmax;
            }
            printf("   [workgroups: first start to last end %.1f ms; per-workgroup %.1f .. %.1f ms] ", (hi - lo) / 1e5, dmin / 1e5, dmax / 1e5);
            const double flop = 262144.0 * iters * (double)wgs * 8;
            printf("%-42s %8.1f ms  %7.1f TFLOP/s   core clock %4.0f MHz   MFMA busy %.2f\n", v.name, ms, flop / ms / 1e9, 100.0 * hclk[0] / hclk[1],
                   flop / (ms * 1e-3) / (1024.0 * 1024.0 * 1e6 * 100.0 * hclk[0] / hclk[1]));
            fflush(stdout);
        }
    return 0;
}
