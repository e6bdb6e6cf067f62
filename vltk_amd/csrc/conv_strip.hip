// 1x1 convolution with small K (Cin = 256 or 512) as a "strip" GEMM: the tile's PIXEL PANEL stays in LDS and the
// workgroup sweeps ALL output channels.
//
// Why: the 1x1 kernels are bound by operand fill through the CU's vector-memory path (~20 B/clk/CU measured, DESIGN.md
// section 6), and a 128 x 256 tile of conv_mfma_duo.hip moves (128 + 256) x 64 B per 32-channel stage: the pixel rows are
// fetched again for every 256-channel block of the output (8 times for Res5 conv3).  Here a workgroup owns 128 pixel rows:
// their whole K extent ([128 x Cin] f16 = 128 KiB at Cin = 512) is brought into LDS ONCE, and only weights stream after
// that: 16 KiB per stage instead of 24 (-33 % bytes per MFMA).
//
//   * 8 waves; wave w owns output channels [w*32, w*32+32) of the current 256-channel block: 128 x 32 per wave,
//     8 x 2 accumulator fragments (64 registers), the same fragment layout and hand-issued ds_read schedule as the other
//     MFMA kernels (16x16x32 f16).
//   * a wave's weight rows are read by nobody else: every wave streams ITS 32 rows through a private two-slot ring
//     (2 x 2 KiB), so the K loop has no workgroup barrier at all -- only the wave's own counted vmcnt / lgkmcnt waits --
//     and the waves de-phase freely (one wave's LDS-DMA issue runs under its SIMD partner's MFMAs).
//   * LDS = S x 8 KiB panel + 8 x 4 KiB weight rings = 160 KiB at S = 16: one workgroup per CU.
//   * epilogue per 256-channel block straight from the accumulators (a lane holds 8 consecutive channels of one
//     pixel row: one 16-byte residual load and one 16-byte store per fragment row); same arithmetic order as the
//     other kernels ((acc + bias) + residual, ReLU, one rounding), so the result is bit-identical.
//
// STATUS: experimental (VK_CONV_STRIP=1 enables it for plain 1x1 layers with Cin 256 / 512; the fused-mean and
// dual-source forms stay on conv_mfma_duo.hip).  Measured on Res5 conv3 (M = 1 881 600, K = 512, N = 2048, residual +
// ReLU), bit-identical to the two-per-CU kernel: 5.7-6.1 ms against 6.1 ms.  In-kernel stamps (wave 0): the K loops run at
// 660-750 core cycles per 32-channel stage (two-per-CU kernel: 1432) -- the fill argument holds -- but every block's
// epilogue takes ~11 000 cycles, as long as the block's sixteen stages: 64 KiB of residual rows + 64 KiB of output per block
// and CU with 64 KiB in flight per CU.  The two phases add up instead of overlapping, and they cannot be interleaved
// inside one wave: residual loads (HBM latency ~5000 cycles = 7 stages) share the in-order vmcnt queue with the wave's
// two-deep weight ring, so a residual request issued more than two stages ahead stalls the weight stream, and a deeper
// ring does not fit beside the 128 KiB panel.  Overlap needs separate waves with their own queues (what the two-per-CU
// kernel does, at 1.5x the fill).  Kept as the measured starting point for that trade-off, not used by default.
#include <cstdio>
#include <type_traits>
#include <vector>

#include "vk_common.h"

namespace vk {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct StripK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int M, cin_bytes, ldy, wrow_bytes, relu, n_tiles;
    int stagger;              // x 8128 cycles of start delay for waves 4-7 (0: none)
    unsigned long *stamps;    // STAMP builds: 5 words per workgroup
};

constexpr int S_BM = 128;
constexpr int S_SLAB = S_BM * 64;        // 8 KiB: one 32-channel stage of the tile's pixel rows
constexpr int S_WSLOT = 32 * 64;         // 2 KiB: one stage of a wave's 32 weight rows
constexpr int S_WWAVE = 2 * S_WSLOT;     // two slots per wave
constexpr int strip_smem(int S) { return S * S_SLAB + 8 * S_WWAVE; }

#define VKS_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int S, bool STAMP>
__global__ __launch_bounds__(512, 2) void conv_strip_kernel(StripK p) {
    constexpr int WBASE = S * S_SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long t0 = 0, t1 = 0, c0 = 0, c1 = 0, cyc_epi = 0, ce0 = 0, ce1 = 0;
    if constexpr (STAMP) t0 = __builtin_amdgcn_s_memrealtime();

    const int m0 = blockIdx.x * S_BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int Q = p.n_tiles * S;                                   // weight stages of the whole sweep

    // ---- LDS-DMA sources: pixel piece = rows wave*16 + (lane>>2) of a slab; weight pieces i = 0,1: rows wave*32 + i*16 + (lane>>2)
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
    const unsigned a_off = (unsigned)min(m0 + wave * 16 + lrow, p.M - 1) * (unsigned)p.cin_bytes + lchunk * 16;
    const unsigned wsrc0 = (unsigned)(wave * 32 + lrow) * (unsigned)p.wrow_bytes + lchunk * 16;
    const unsigned wstep = 16u * p.wrow_bytes, wtile = 256u * p.wrow_bytes;
    auto req_x = [&](int s) { VKS_GLDS16(p.x + (a_off + (unsigned)s * 64u), smem + s * S_SLAB + wave * 1024); };
    auto req_w = [&](int q, int i) {
        unsigned a = wsrc0;
        asm volatile("" : "+v"(a));     // opaque: keeps the add in the loop instead of a register per (stage, piece)
        const unsigned nt = (unsigned)q / (unsigned)S, ks = (unsigned)q % (unsigned)S;
        VKS_GLDS16(p.w + (a + nt * wtile + i * wstep + ks * 64u), smem + WBASE + wave * S_WWAVE + (q & 1) * S_WSLOT + i * 1024);
    };

    // ---- fragment read addresses ----
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned x_a = lds0 + j * 64 + ((g ^ ((-(j >> 2)) & 3)) << 4);                 // + slab*8192 + mi*1024
    unsigned w_a[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int wrow = (j >> 2) * 8 + par * 4 + (j & 3);                               // row within the wave's 32
        w_a[par] = lds0 + WBASE + wave * S_WWAVE + wrow * 64 + ((g ^ ((-(wrow >> 2)) & 3)) << 4);
    }

    floatx4 acc[8][2];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 wa[2], wb[2], xw[4];

#define VKS_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define VKS_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg))
#define VKS_SB() __builtin_amdgcn_sched_barrier(0)
#define VKS_MMA_ROW(MI, XR, WF)                                                                      \
    do {                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[MI][ni] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[ni], XR, acc[MI][ni], 0, 0, 0);                \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)

    // PRE(q): rows 0-4 of stage q; then every fragment of the stage is in registers (lgkmcnt(0)), the wave's weight slot
    // q & 1 is free: stage q+2 is requested into it, and the wave waits for ITS pieces of stage q+1 (all but the two just
    // issued).  POST(q): rows 5-7 with the first reads of stage q+1.  No read is in flight at any branch.
    auto pre = [&](int q, const half8 (&wcur)[2]) {
        const unsigned xs = x_a + (unsigned)(q % S) * S_SLAB;
        VKS_DSR(xw[3], xs, 3072); VKS_WAIT3(xw[0]); VKS_SB(); VKS_MMA_ROW(0, xw[0], wcur); VKS_SB();
        VKS_DSR(xw[0], xs, 4096); VKS_WAIT3(xw[1]); VKS_SB(); VKS_MMA_ROW(1, xw[1], wcur); VKS_SB();
        VKS_DSR(xw[1], xs, 5120); VKS_WAIT3(xw[2]); VKS_SB(); VKS_MMA_ROW(2, xw[2], wcur); VKS_SB();
        VKS_DSR(xw[2], xs, 6144); VKS_WAIT3(xw[3]); VKS_SB(); VKS_MMA_ROW(3, xw[3], wcur); VKS_SB();
        VKS_DSR(xw[3], xs, 7168); VKS_WAIT3(xw[0]); VKS_SB(); VKS_MMA_ROW(4, xw[0], wcur); VKS_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
        VKS_SB();
        if (q + 2 < Q) {
            req_w(q + 2, 0);
            req_w(q + 2, 1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        VKS_SB();
    };
    auto post = [&](int q, const half8 (&wcur)[2], half8 (&wnext)[2]) {
        const unsigned xn = x_a + (unsigned)((q + 1) % S) * S_SLAB, sn = (unsigned)((q + 1) & 1) * S_WSLOT;
        VKS_DSR(wnext[0], w_a[0] + sn, 0);
        VKS_DSR(wnext[1], w_a[1] + sn, 0);
        VKS_DSR(xw[0], xn, 0);
        VKS_SB();
        VKS_MMA_ROW(5, xw[1], wcur);
        VKS_SB();
        VKS_DSR(xw[1], xn, 1024);
        VKS_SB();
        VKS_MMA_ROW(6, xw[2], wcur);
        VKS_SB();
        VKS_DSR(xw[2], xn, 2048);
        VKS_SB();
        VKS_MMA_ROW(7, xw[3], wcur);
        VKS_SB();
    };
    auto last_rows = [&](const half8 (&wcur)[2]) {
        VKS_MMA_ROW(5, xw[1], wcur);
        VKS_MMA_ROW(6, xw[2], wcur);
        VKS_MMA_ROW(7, xw[3], wcur);
    };

    // ---- prologue: the whole pixel panel (one piece per wave and slab) and the wave's weights of stages 0 and 1 ----
#pragma unroll
    for (int s = 0; s < S; ++s) req_x(s);
    req_w(0, 0);
    req_w(0, 1);
    if (Q > 1) {
        req_w(1, 0);
        req_w(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");          // the panel's pieces come from all eight waves
    if constexpr (STAMP) {
        t1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    }

    const int col = wave * 32 + g * 8;               // tile-local channel of this lane's 8 values
    // Experiment (VK_STRIP_STAGGER=n, default 0): the sweep has no barrier and the two waves of a SIMD (w and w + 4) have
    // identical work, so they stay in phase; a one-time delay of n x 8128 cycles for waves 4-7 would put one wave's epilogue
    // under its partner's K loop.  Measured: no systematic effect (5.69-6.10 ms with and without, same box).
    if (p.stagger && wave >= 4) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int nt = 0; nt < p.n_tiles; ++nt) {
        const int q0 = nt * S;
        VKS_DSR(wa[0], w_a[0] + (unsigned)(q0 & 1) * S_WSLOT, 0);
        VKS_DSR(wa[1], w_a[1] + (unsigned)(q0 & 1) * S_WSLOT, 0);
        VKS_DSR(xw[0], x_a, 0);
        VKS_DSR(xw[1], x_a, 1024);
        VKS_DSR(xw[2], x_a, 2048);
        pre(q0, wa);
#pragma unroll 1
        for (int s = 0; s + 2 < S; s += 2) {
            post(q0 + s, wa, wb);
            pre(q0 + s + 1, wb);
            post(q0 + s + 1, wb, wa);
            pre(q0 + s + 2, wa);
        }
        post(q0 + S - 2, wa, wb);
        pre(q0 + S - 1, wb);
        last_rows(wb);

        // ---- epilogue of this 256-channel block, straight from the accumulators ----
        if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ce0)::"memory");
        const int n0 = nt * 256;
        const floatx4 b0 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[0];
        const floatx4 b1 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[1];
        half8 rr[8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = min(m0 + mi * 16 + j, p.M - 1);
            if (p.res)
                rr[mi] = *reinterpret_cast<const half8 *>(p.res + ((long)m * p.ldy + n0 + col) * 2);
            else
                rr[mi] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + mi * 16 + j;
            const floatx4 v0 = acc[mi][0] + b0, v1 = acc[mi][1] + b1;
            half8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = v0[e] + (float)rr[mi][e], b = v1[e] + (float)rr[mi][4 + e];
                if (p.relu) {
                    a = a > 0.f ? a : 0.f;
                    b = b > 0.f ? b : 0.f;
                }
                o[e] = (_Float16)a;
                o[4 + e] = (_Float16)b;
            }
            if (m < p.M) *reinterpret_cast<half8 *>(p.y + ((long)m * p.ldy + n0 + col) * 2) = o;
            acc[mi][0] = floatx4{0.f, 0.f, 0.f, 0.f};
            acc[mi][1] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (STAMP) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ce1)::"memory");
            cyc_epi += ce1 - ce0;
        }
    }
#undef VKS_DSR
#undef VKS_WAIT3
#undef VKS_MMA_ROW
#undef VKS_SB
    if constexpr (STAMP) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long t2 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long *o = p.stamps + (long)blockIdx.x * 5;
            o[0] = t1 - t0;          // prologue, 100 MHz ticks
            o[1] = t2 - t1;          // sweep + epilogues
            o[2] = c1 - c0;          // the same in core-clock cycles
            o[3] = Q;
            o[4] = cyc_epi;          // of which: in the epilogues (wave 0)
        }
    }
}

bool conv_strip_eligible(const ConvArgs &a) {
    const char *v = getenv("VK_CONV_STRIP");             // experimental: "1" enables it where it is legal
    if (!v || v[0] != '1') return false;
    if (a.stem || a.dt != VK_F16 || a.out_dt != VK_F16 || a.x2 || a.pool_part || a.groups > 1) return false;
    if (a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride != 1 || a.relu > 1) return false;
    if (a.Cout % 256 != 0 || a.ldy != a.Cout || (a.Cin != 256 && a.Cin != 512)) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (M < 8 * S_BM || M * a.Cin * 2 >= (1L << 32) || (long)a.Cout * a.Cin * 2 >= (1L << 32)) return false;
    return true;
}

int launch_conv_strip(const ConvArgs &a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_strip_kernel<16, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, strip_smem(16)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_strip_kernel<8, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, strip_smem(8)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_strip_kernel<16, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, strip_smem(16)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_strip_kernel<8, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, strip_smem(8)));
        attr_set = true;
    }
    StripK k;
    k.x = (const char *)a.x;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    const long M = (long)a.N * a.Ho * a.Wo;
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.ldy = a.ldy;
    k.wrow_bytes = a.Cin * 2;
    k.relu = a.relu;
    k.n_tiles = a.Cout / 256;
    k.stamps = nullptr;
    k.stagger = getenv("VK_STRIP_STAGGER") ? atoi(getenv("VK_STRIP_STAGGER")) : 0;
    const int S = a.Cin / 32;
    const dim3 grid(ceil_div(k.M, S_BM)), block(512);
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    if (const char *sf = getenv("VK_STRIP_STAMPS")) {    // diagnostic: one launch, per-workgroup stamps appended to the file
        const size_t nb = (size_t)grid.x * 5 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        if (S == 16)
            hipLaunchKernelGGL((conv_strip_kernel<16, true>), grid, block, strip_smem(16), stream, k);
        else
            hipLaunchKernelGGL((conv_strip_kernel<8, true>), grid, block, strip_smem(8), stream, k);
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> h((size_t)grid.x * 5);
        VK_CHECK_HIP(hipMemcpy(h.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# launch M=%d cout=%d cin=%d grid=%u\n", k.M, a.Cout, a.Cin, grid.x);
            for (unsigned b = 0; b < grid.x; ++b) fprintf(f, "%u %lu %lu %lu %lu %lu\n", b, h[b * 5], h[b * 5 + 1], h[b * 5 + 2], h[b * 5 + 3], h[b * 5 + 4]);
            fclose(f);
        }
    } else if (S == 16)
        hipLaunchKernelGGL((conv_strip_kernel<16, false>), grid, block, strip_smem(16), stream, k);
    else
        hipLaunchKernelGGL((conv_strip_kernel<8, false>), grid, block, strip_smem(8), stream, k);
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        tm->recs.push_back({a.concurrent ? 6 : 5, 2.0 * (double)k.M * a.Cout * a.Cin, e0, e1, k.M, a.Cout, a.Cin, 1, 1,
                            2.0 * ((double)k.M * a.Cin + (double)k.M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * a.Cin)});
    }
    return VK_OK;
}

}  // namespace vk
