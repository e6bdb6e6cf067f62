// Variant "B" of the 256x256 LDS-ring convolution (see conv_mfma256.hip for the contract and the
// common structure): K advances in stages of 64 channels held as 128-BYTE LDS rows, so that every
// LDS-DMA wave-instruction moves 8 rows x 128 B = 8 WHOLE cache lines (variant A's 32-channel stages
// move 16 half lines per instruction, which doubles the texture-addresser work per byte).
//
//   * LDS = 2 slots x (256 pixel rows + 256 channel rows) x 128 B = 128 KiB.
//   * A stage is consumed as two half-stages (channels 0-31 / 32-63 of the slot), each 8 row-steps of
//     4 MFMAs with its own weight-fragment set; pixel-row fragments stream through the same 4-deep
//     register window (3 row-steps ahead), straight across half-stage and stage boundaries.
//   * ONE barrier per stage (64 MFMAs per wave), in the second half-stage after the last fragment
//     read of the slot has been issued: vmcnt(0) (stage s+1 has landed) + lgkmcnt(0) + s_barrier,
//     then the 8 DMA pieces of stage s+2 go out two per row-step behind MFMA groups.
//   * swizzles (16-B chunk c of row r is stored at c ^ key): pixel rows key = r & 7; weight rows are
//     read in the interleaved order that gives a lane 8 consecutive output channels, for which
//     key = (r & 7) ^ (((r >> 3) & 3) << 1) makes every ds_read_b128 lane group hit 16 distinct slots.
#include <type_traits>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct Conv256BK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    const char *zero;
    int H, W, Ho, Wo, HoWo, M;
    int cin_bytes, ldy;
    int kw, stride, pad, dil;
    int stages, st_per_tap;   // K stages of 64 channels
    int wrow_bytes;
    int relu;
    int m_tiles, n_tiles;
};

constexpr int B_BM = 256, B_BN = 256, B_ROWB = 128;
constexpr int B_XB = B_BM * B_ROWB;          // 32 KiB
constexpr int B_SLOT = 2 * B_XB;             // 64 KiB
constexpr int B_SMEM = 2 * B_SLOT;           // 128 KiB

#define VKB_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int DBG>
__global__ __launch_bounds__(512, 2) void conv_mfma256b_kernel(Conv256BK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tile = t % p.n_tiles, m_tile = t / p.n_tiles;
    const int m0 = m_tile * B_BM, n0 = n_tile * B_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, j = lane & 15;

    // ---- LDS-DMA source state: this lane feeds rows (wave*4+i)*8 + (lane>>3), i = 0..3, of both operands ----
    const int row8 = lane >> 3;
    const int xchunk = (lane & 7) ^ row8;                  // logical chunk landing at phys chunk lane&7 (pixel rows)
    int off16[4];                                          // (n, h0, w0, chunk) byte offset / 16, signed
    int hw[4];                                             // h0 (low 16) | w0 (high 16), signed; invalid row -> h0 = -32768
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + row8;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n_img = mm / p.HoWo;
        const int rem = mm - n_img * p.HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int h0 = ok ? ho * p.stride - p.pad : -32768, w0 = wo * p.stride - p.pad;
        off16[i] = (int)((((long)(n_img * p.H + (ok ? h0 : 0)) * p.W + w0) * p.cin_bytes) >> 4) + xchunk;
        hw[i] = (h0 & 0xffff) | (w0 << 16);
    }
    const char *wsrc0 = p.w + (long)(n0 + wave * 32 + row8) * p.wrow_bytes;   // weight row of piece i: + i*8 rows
    const long wstep = 8L * p.wrow_bytes;
    const int dma0 = wave * 4 * 1024;                      // this wave's first row block inside an operand's slot image

    int khi = 0, kwi = 0, kc = 0;   // tap / channel-stage of the NEXT pixel request
    int cur_dh = 0, cur_dw = 0;     // tap offsets of the request being issued (wave-uniform)
    long cur_toff = 0;
    auto prep_x = [&]() {           // scalar part: tap offsets of the next stage, then advance the tap state
        cur_dh = khi * p.dil;
        cur_dw = kwi * p.dil;
        cur_toff = ((long)cur_dh * p.W + cur_dw) * p.cin_bytes + kc * B_ROWB;
        kc += 1;
        const int c1 = (kc == p.st_per_tap) ? 1 : 0;
        kc *= (1 - c1);
        kwi += c1;
        const int c2 = (kwi == p.kw) ? 1 : 0;
        kwi *= (1 - c2);
        khi += c2;
    };
    auto req_x = [&](int stage, int i) {   // per-lane source address computed right at the issue (short live range)
        if constexpr (DBG & 1) return;
        int o16 = off16[i], hwv = hw[i];
        asm volatile("" : "+v"(o16), "+v"(hwv));   // opaque: stops LICM from keeping 4 hoisted 64-bit addresses live
        const int h0 = (int)(short)(hwv & 0xffff), w0 = hwv >> 16;
        const bool ok = (unsigned)(h0 + cur_dh) < (unsigned)p.H && (unsigned)(w0 + cur_dw) < (unsigned)p.W;
        const char *src = (ok && !(DBG & 4)) ? p.x + ((long)o16 << 4) + cur_toff : p.zero;   // DBG 4: all DMA reads hit one line
        VKB_GLDS16(src, smem + (stage & 1) * B_SLOT + dma0 + i * 1024);
    };
    auto req_w = [&](int stage, int i) {
        if constexpr (DBG & 1) return;
        // weight rows use key = (r&7) ^ (((r>>3)&3)<<1); r = (wave*4+i)*8 + row8  =>  key = row8 ^ ((i&3)<<1)
        int lchunk = (lane & 7) ^ row8 ^ ((i & 3) << 1);
        asm volatile("" : "+v"(lchunk));           // opaque (see req_x)
        VKB_GLDS16((DBG & 4) ? p.zero : wsrc0 + i * wstep + (long)stage * B_ROWB + lchunk * 16,
                   smem + (stage & 1) * B_SLOT + B_XB + dma0 + i * 1024);
    };

    // ---- fragment read addresses (LDS byte addresses, slot 0, half-stage kh) ----
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    // (the second channel half is chunk + 4, i.e. byte address ^ 0x40: derived on the fly, not kept in registers;
    //  slot bases are multiples of 64 KiB and lds0 is 16-byte aligned with bits 4-6 of every row start clear)
    unsigned x_a0, w_a0[2];
    x_a0 = (wr * 128 + j) * B_ROWB + ((g ^ (j & 7)) << 4);                                   // + mi*2048
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int wrow = wc * 64 + (j >> 2) * 8 + par * 4 + (j & 3);                          // + (ni>>1)*32 rows
        const int key = (wrow & 7) ^ (((wrow >> 3) & 3) << 1);
        w_a0[par] = B_XB + wrow * B_ROWB + ((g ^ key) << 4);
    }
    auto xaddr = [&](int kh, unsigned slot) { return lds0 + ((x_a0 ^ (kh << 6)) + slot); };
    auto waddr = [&](int par, int kh, unsigned slot) { return lds0 + ((w_a0[par] ^ (kh << 6)) + slot); };

    floatx4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};

    half8 wa[4], wb[4], xw[4];
    const int S = p.stages;

#define VKB_DSR(dst, addr, OFF)                                                               \
    do {                                                                                      \
        if constexpr (DBG & 2)                                                                \
            asm volatile("" : "+v"(dst) : "v"(addr));                                         \
        else                                                                                  \
            asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr));         \
    } while (0)
#define VKB_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg))
#define VKB_MMA_ROW(MI, XR, WF)                                                                      \
    do {                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[MI][ni] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[ni], XR, acc[MI][ni], 0, 0, 0);                \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)
#define VKB_READ_W(WF, a0, a1)    \
    VKB_DSR(WF[0], a0, 0);        \
    VKB_DSR(WF[1], a1, 0);        \
    VKB_DSR(WF[2], a0, 4096);     \
    VKB_DSR(WF[3], a1, 4096)
#define VKB_SB() __builtin_amdgcn_sched_barrier(0)

    // One half-stage = 8 row-steps on channels [32*KH, 32*KH+32) of stage s.  Rows 0-4 read row r+3 of the
    // same half-stage; rows 5-7 read rows 0-2 of the NEXT half-stage (same slot for KH=0, the other slot --
    // after the barrier -- for KH=1).  FULL: steady state (stage s+2 exists), no run-time guards.
    auto half_body = [&](auto full_c, auto kh_c, int s, const half8 (&wcur)[4], half8 (&wnext)[4]) {
        constexpr bool FULL = decltype(full_c)::value;
        constexpr int KH = decltype(kh_c)::value;
        const unsigned so = (unsigned)(s & 1) * B_SLOT, sn = (unsigned)((s + 1) & 1) * B_SLOT;
        const unsigned xs = xaddr(KH, so);
        const bool more = KH == 0 || FULL || (s + 1 < S);          // is there a next half-stage?
        const bool rq = KH == 1 && (FULL || (s + 2 < S));          // does this half-stage launch the DMA of stage s+2?
        const bool rq0 = KH == 0 && (FULL || (s + 1 < S)) && s > 0;   // tail pieces of stage s+1's request (rows 0-1)
        // pieces of request(stage s+1) still to issue at the start of a KH=0 half: the 4 weight pieces (2 per row)
        VKB_DSR(xw[3], xs, 6144); VKB_WAIT3(xw[0]); VKB_SB(); VKB_MMA_ROW(0, xw[0], wcur); VKB_SB();
        if (rq0) { req_w(s + 1, 0); req_w(s + 1, 1); }
        VKB_SB();
        VKB_DSR(xw[0], xs, 8192); VKB_WAIT3(xw[1]); VKB_SB(); VKB_MMA_ROW(1, xw[1], wcur); VKB_SB();
        if (rq0) { req_w(s + 1, 2); req_w(s + 1, 3); }
        VKB_SB();
        VKB_DSR(xw[1], xs, 10240); VKB_WAIT3(xw[2]); VKB_SB(); VKB_MMA_ROW(2, xw[2], wcur); VKB_SB();
        if (rq) prep_x();
        VKB_DSR(xw[2], xs, 12288); VKB_WAIT3(xw[3]); VKB_SB(); VKB_MMA_ROW(3, xw[3], wcur); VKB_SB();
        VKB_DSR(xw[3], xs, 14336); VKB_WAIT3(xw[0]); VKB_SB(); VKB_MMA_ROW(4, xw[0], wcur); VKB_SB();
        if (more) {
            unsigned xn, wn0, wn1;
            if constexpr (KH == 0) {
                // same slot, second channel half: the data is already there, no synchronisation needed
                xn = xaddr(1, so);
                wn0 = waddr(0, 1, so);
                wn1 = waddr(1, 1, so);
            } else {
                // every fragment read of slot(s) has been issued: once complete (lgkmcnt(0)) and every wave is
                // here the slot may be refilled; vmcnt(0): this wave's pieces of stage s+1 have landed, and after
                // the barrier everybody's have
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
                VKB_SB();
                xn = xaddr(0, sn);
                wn0 = waddr(0, 0, sn);
                wn1 = waddr(1, 0, sn);
            }
            VKB_READ_W(wnext, wn0, wn1);
            VKB_DSR(xw[0], xn, 0);
            VKB_SB();
            VKB_MMA_ROW(5, xw[1], wcur);
            VKB_SB();
            if (rq) { req_x(s + 2, 0); req_x(s + 2, 1); }
            VKB_DSR(xw[1], xn, 2048);
            VKB_SB();
            VKB_MMA_ROW(6, xw[2], wcur);
            VKB_SB();
            if (rq) { req_x(s + 2, 2); req_x(s + 2, 3); }
            VKB_DSR(xw[2], xn, 4096);
            VKB_SB();
            VKB_MMA_ROW(7, xw[3], wcur);
            VKB_SB();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3]));
            VKB_SB();
            VKB_MMA_ROW(5, xw[1], wcur);
            VKB_MMA_ROW(6, xw[2], wcur);
            VKB_MMA_ROW(7, xw[3], wcur);
        }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;

    // prologue: stage 0 and stage 1 completely (8 pieces each); stage 0 must have landed
    prep_x();
#pragma unroll
    for (int i = 0; i < 4; ++i) req_x(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) req_w(0, i);
    if (S > 1) {
        prep_x();
#pragma unroll
        for (int i = 0; i < 4; ++i) req_x(1, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) req_w(1, i);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    {
        const unsigned a0 = waddr(0, 0, 0u), a1 = waddr(1, 0, 0u), ax = xaddr(0, 0u);
        VKB_READ_W(wa, a0, a1);
        VKB_DSR(xw[0], ax, 0);
        VKB_DSR(xw[1], ax, 2048);
        VKB_DSR(xw[2], ax, 4096);
    }
    // Request bookkeeping: stage s+2's pixel pieces are issued in the second half of stage s (rows 5-6, after
    // the barrier) and its weight pieces in the first half of stage s+1 (rows 0-1): that is `rq0` with s := s+1,
    // i.e. "the weight pieces of stage s+1" in a KH=0 half with s > 0... see half_body.  Stage 1's weight
    // pieces were issued in the prologue, hence the s > 0 guard there.
    int s = 0;
    for (; s + 2 < S; ++s) {
        half_body(T_{}, K0{}, s, wa, wb);
        half_body(T_{}, K1{}, s, wb, wa);
    }
    for (; s < S; ++s) {
        half_body(F_{}, K0{}, s, wa, wb);
        half_body(F_{}, K1{}, s, wb, wa);
    }
#undef VKB_DSR
#undef VKB_WAIT3
#undef VKB_MMA_ROW
#undef VKB_READ_W
#undef VKB_SB

    // ---- epilogue (identical to variant A): through LDS, whole 512-B rows to/from HBM ----
    asm volatile("s_barrier" ::: "memory");
    floatx4 *stg = reinterpret_cast<floatx4 *>(smem);
    auto load_res = [&](int h, half8 (&rr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int it = tid + 512 * i, row = it >> 5, k8 = it & 31;
            const int m = min(m0 + h * 128 + row, p.M - 1);
            if (p.res)
                rr[i] = *reinterpret_cast<const half8 *>(p.res + ((long)m * p.ldy + n0 + k8 * 8) * 2);
            else
                rr[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto stage_half = [&](int h) {
        if (wr == h) {
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                const int col = wc * 64 + qn * 32 + g * 8;
                const floatx4 b0 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[0];
                const floatx4 b1 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[1];
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int row = mi * 16 + j;
                    const int c16 = col >> 2;
                    stg[row * 64 + (c16 ^ (row & 7))] = acc[mi][2 * qn] + b0;
                    stg[row * 64 + ((c16 + 1) ^ (row & 7))] = acc[mi][2 * qn + 1] + b1;
                }
            }
        }
    };
    auto write_half = [&](int h, const half8 (&rr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int it = tid + 512 * i, row = it >> 5, k8 = it & 31;
            const int m = m0 + h * 128 + row;
            const floatx4 v0 = stg[row * 64 + ((2 * k8) ^ (row & 7))];
            const floatx4 v1 = stg[row * 64 + ((2 * k8 + 1) ^ (row & 7))];
            half8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = v0[e] + (float)rr[i][e], b = v1[e] + (float)rr[i][4 + e];
                if (p.relu) {
                    a = a > 0.f ? a : 0.f;
                    b = b > 0.f ? b : 0.f;
                }
                o[e] = (_Float16)a;
                o[4 + e] = (_Float16)b;
            }
            if (m < p.M) *reinterpret_cast<half8 *>(p.y + ((long)m * p.ldy + n0 + k8 * 8) * 2) = o;
        }
    };
#define VKB_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    half8 r0[8];
    load_res(0, r0);
    stage_half(0);
    VKB_LDS_BARRIER();
    write_half(0, r0);
    load_res(1, r0);
    VKB_LDS_BARRIER();
    stage_half(1);
    VKB_LDS_BARRIER();
    write_half(1, r0);
#undef VKB_LDS_BARRIER
}

bool conv256b_eligible(const ConvArgs &a) {
    if (a.stem || a.dt != VK_F16 || a.out_dt != VK_F16) return false;
    if (a.Cout % B_BN != 0 || a.ldy != a.Cout) return false;
    if (a.Cin % 64 != 0) return false;
    if ((long)a.N * a.H * a.W * a.Cin * 2 >= (1L << 35)) return false;   // 16-byte-unit offsets are 32-bit signed
    return (long)a.N * a.Ho * a.Wo >= 4 * B_BM;
}

int launch_conv256b(const ConvArgs &a, hipStream_t stream) {
    static char *zero_page = nullptr;
    if (!zero_page) {
        VK_CHECK_HIP(hipMalloc((void **)&zero_page, 256));
        VK_CHECK_HIP(hipMemset(zero_page, 0, 256));
    }
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256b_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256b_kernel<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256b_kernel<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256b_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256b_kernel<4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM));
        attr_set = true;
    }
    Conv256BK k;
    k.x = (const char *)a.x;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    k.zero = zero_page;
    k.H = a.H;
    k.W = a.W;
    k.Ho = a.Ho;
    k.Wo = a.Wo;
    k.HoWo = a.Ho * a.Wo;
    const long M = (long)a.N * a.Ho * a.Wo;
    VK_REQUIRE(M > 0 && M < (1L << 31) - B_BM, VK_EINVAL, "conv256b: M=%ld out of range", M);
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.ldy = a.ldy;
    k.kw = a.kw;
    k.stride = a.stride;
    k.pad = a.pad;
    k.dil = a.dil;
    k.st_per_tap = a.Cin / 64;
    k.stages = a.kh * a.kw * k.st_per_tap;
    k.wrow_bytes = a.kh * a.kw * a.Cin * 2;
    k.relu = a.relu;
    k.m_tiles = ceil_div(k.M, B_BM);
    k.n_tiles = a.Cout / B_BN;
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    const int dbg = getenv("VK_CONV256_DBG") ? atoi(getenv("VK_CONV256_DBG")) & 7 : 0;
    const dim3 grid(k.m_tiles * k.n_tiles), block(512);
    switch (dbg) {
        case 1: hipLaunchKernelGGL(conv_mfma256b_kernel<1>, grid, block, B_SMEM, stream, k); break;
        case 2: hipLaunchKernelGGL(conv_mfma256b_kernel<2>, grid, block, B_SMEM, stream, k); break;
        case 3: hipLaunchKernelGGL(conv_mfma256b_kernel<3>, grid, block, B_SMEM, stream, k); break;
        case 4: hipLaunchKernelGGL(conv_mfma256b_kernel<4>, grid, block, B_SMEM, stream, k); break;
        default: hipLaunchKernelGGL(conv_mfma256b_kernel<0>, grid, block, B_SMEM, stream, k);
    }
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        tm->recs.push_back({0, 2.0 * (double)k.M * a.Cout * a.kh * a.kw * a.Cin, e0, e1, k.M, a.Cout, a.Cin, a.kh * a.kw, a.stride,
                            2.0 * ((double)a.N * a.H * a.W * a.Cin + (double)k.M * a.Cout * (a.res ? 2 : 1) +
                                   (double)a.Cout * a.kh * a.kw * a.Cin)});
    }
    return VK_OK;
}

}  // namespace vk
