// 1x1 convolution (any stride, no padding) as a 128x256-tile GEMM with TWO co-resident workgroups per CU.
//
// Why: the 256x256 ring kernel (conv_mfma256.hip) owns the CU, so a tile's epilogue (residual read +
// output write: 256 KiB of HBM traffic, ~13 us at the CU's fair share of HBM when every CU does it at
// once) cannot overlap anything: for the small-K 1x1 layers (Res5 conv3: K = 512) the memory phase is as
// long as the MFMA phase and the two add up.  Here a workgroup is 4 waves (one per
// SIMD, each 128 pixels x 64 channels: the same per-wave geometry, fragment layout and hand-issued
// ds_read schedule as the ring kernel) with a 72 KiB LDS ring, so two workgroups share a CU and one's
// epilogue runs under the other's MFMA loop.  (The two de-phase by themselves: in-kernel stamps show ~90 % of
// epilogue time under the co-resident workgroup's MFMA loop; a forced start skew changed nothing and was removed,
// and so were non-temporal residual loads / output stores: 0...-1 % on every shape.)
//
//   * tile 128 pixels x 256 channels; K stages of 32 channels (64-B LDS rows, XOR swizzle on the DMA source);
//     ring of 3 slots x (8 KiB pixels + 16 KiB weights); per stage a wave issues 2 pixel + 4 weight pieces
//     (global_load_lds_dwordx4, saddr form: 32-bit per-lane offsets from the uniform tensor base).
//   * stage s: rows 0-2 carry the last three weight pieces of stage s+2; the barrier sits after row 4
//     (vmcnt(6): stage s+1 landed, only the six pieces of stage s+2 may be outstanding); rows 5-7 carry the
//     two pixel pieces and the first weight piece of stage s+3 into the slot the barrier just freed.
//   * rows >= M are clamped to row M-1 for the loads and masked at the store (no zero page: a 1x1 conv has
//     no padded taps).
//   * epilogue through LDS in two 64-row halves ([64][256] f32 = 64 KiB), whole 512-B rows to HBM.
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "vk_common.h"

namespace vk {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// element type of the operands / output: f16 (detector) or bf16 (the encoder's GEMMs); accumulation is fp32 either way
template <typename ET>
struct DuoT;
template <>
struct DuoT<_Float16> {
    typedef half8 vec;
    static __device__ __forceinline__ floatx4 mfma(vec a, vec b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct DuoT<__bf16> {
    typedef bf16x8 vec;
    static __device__ __forceinline__ floatx4 mfma(vec a, vec b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

struct DuoK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int H, W, Ho, Wo, HoWo, M;
    int cin_bytes, ldy;
    int stride;
    const char *x2;           // second K segment (stages >= split): pixel rows of [M, cin2], stride 1; or nullptr
    int cin2_bytes;
    int split;                // stages of the first segment (== stages when x2 is null)
    int stages;               // (Cin + Cin2) / 32
    int wrow_bytes;           // Cin * 2
    int relu;
    int m_tiles, n_tiles;
    int *pool_part;           // [m_tiles][2][Cout][hi, lo]: exact integer column sums of the tile's rows, split at the
                              // image boundary (y is not written)
    unsigned long *stamps;    // STAMP builds only: 12 words per workgroup (phase times, HW_ID, XCC_ID, cycle sums)
};

constexpr int D_BM = 128, D_BN = 256, D_NSLOT = 3;
constexpr int D_XB = D_BM * 64;              // 8 KiB: one stage of pixel rows
constexpr int D_WB = D_BN * 64;              // 16 KiB: one stage of weight rows
// LDS = [pixel ring: XS slots][weight ring: 3 slots].  XS = 4 (80 KiB each, two workgroups fill the CU's 160 KiB
// exactly) requests pixel rows one stage earlier than weights; measured identical to XS = 3 on every shape
// (interleaved runs, one device), and in-kernel cycle stamps agree: a wave waits for DMA only 3.5-7.5 % of its
// MFMA loop (tools/duo_stamps.py), so the product uses XS = 3.
constexpr int d_smem(int XS) { return XS * D_XB + D_NSLOT * D_WB; }

#define VKD_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// STAMP: diagnostic build (VK_DUO_STAMPS=<file>): wave 0 records s_memrealtime at the phase boundaries into a
// buffer nothing else reads (cdna guide section 7, in-kernel stamps); never used by the product path.
// DBG: ablation bit mask (timing-only STAMP builds, WRONG results; VK_DUO_DBG with VK_DUO_STAMPS): 1 no weight DMA, 2 no pixel
// DMA after the prologue, 4 no fragment ds_reads, 8 no MFMAs.  Core cycles per 32-MFMA stage on Res5 conv3 (K = 512, one
// wave, partner workgroup mostly in its epilogue): full 1432 | no weight DMA 1217 | no DMA 1092 | no ds_reads 1316 | MFMA
// only 938 | no MFMA 1272 | ds_reads only 650 | DMA only 1240.  The six LDS-DMA pieces alone take 1240 cycles: 24 KiB per
// stage and workgroup = 20 B/clk/CU, plus the partner's epilogue traffic on the same vector-memory path, against the
// ~33 B/clk/CU the L2 -> LDS path delivers at best (MI355X_MICROARCH.md, gather into LDS): a 128 x 256 tile is bound by
// operand fill ((128 + 256) x 64 B per stage), not by MFMA issue.  An eight-wave build (two waves per SIMD and
// workgroup, 128 x 32 per wave, <= 128 VGPRs) was correct and SLOWER (941 vs 760 us: 1812 cycles per stage, 28 % at the
// barrier): more issuing waves do not raise the fill rate.  Weights as compiler-tracked global_load_dwordx4 into
// registers were slower too (1869 cycles on Res5 conv1), so the pieces stay DMA
// GELU: the erf-form GELU epilogue (encoder FFN) is a separate instantiation: sixteen inlined erff() in the epilogue of
// every build cost the detector 16 % (instruction footprint), measured
// TAG 1: the same code under a second symbol for launches of the two-stream backbone section, so that profilers list the
// launches that overlap another stream apart from the ones that run alone
template <typename ET, bool STAMP, int XS, int DBG = 0, bool GELU = false, int TAG = 0>
__global__ __launch_bounds__(256, 2) void conv_duo_kernel(DuoK p) {
    typedef typename DuoT<ET>::vec vec8;
    constexpr int D_WBASE = XS * D_XB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long ts[5], cyc_vm = 0, cyc_bar = 0, cyc0 = 0, cyc1 = 0;
    if constexpr (STAMP) ts[0] = __builtin_amdgcn_s_memrealtime();

    // XCD-aware (bijective) workgroup -> tile map: consecutive tiles (same pixel rows, next channel block)
    // stay on one XCD's L2
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tile = t % p.n_tiles, m_tile = t / p.n_tiles;
    const int m0 = m_tile * D_BM, n0 = n_tile * D_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = channel block of 64
    const int g = lane >> 4, j = lane & 15;

    // ---- LDS-DMA source state: pixel rows (wave*2+i)*16 + (lane>>2), weight rows (wave*4+i)*16 + (lane>>2) ----
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
    unsigned a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + (wave * 2 + i) * 16 + lrow, p.M - 1);
        const int n_img = m / p.HoWo;
        const int rem = m - n_img * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_off[i] = (unsigned)((n_img * p.H + ho * p.stride) * p.W + wo * p.stride) * (unsigned)p.cin_bytes + lchunk * 16;
    }
    // dual-source form (conv3 + projection shortcut as ONE GEMM, K = [conv3 input | block input]): the second
    // segment's rows are the same pixels of another tensor
    unsigned a2_off[2] = {0u, 0u};
    if (p.x2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            a2_off[i] = (unsigned)min(m0 + (wave * 2 + i) * 16 + lrow, p.M - 1) * (unsigned)p.cin2_bytes + lchunk * 16;
    }
    const unsigned wsrc0 = (unsigned)(n0 + wave * 64 + lrow) * (unsigned)p.wrow_bytes + lchunk * 16;
    const unsigned wstep = 16u * p.wrow_bytes;
    auto req_x = [&](int stage, int i) {                             // pixel ring slot = stage % XS
        const int slot = XS == 4 ? (stage & 3) : stage % XS;
        const bool second = stage >= p.split;                        // uniform
        unsigned a = second ? a2_off[i] : a_off[i];
        asm volatile("" : "+v"(a));     // opaque: keeps the add in the loop instead of a register per (stage, piece)
        if constexpr ((DBG & 2) != 0)
            if (stage >= XS) return;                                 // ablation: no pixel DMA after the prologue
        const char *base = second ? p.x2 : p.x;
        VKD_GLDS16(base + (a + (unsigned)(stage - (second ? p.split : 0)) * 64u), smem + slot * D_XB + (wave * 2 + i) * 1024);
    };
    auto req_w = [&](int stage, int slot, int i) {
        unsigned a = wsrc0;
        asm volatile("" : "+v"(a));
        if constexpr ((DBG & 1) != 0) return;                        // ablation: no weight DMA
        VKD_GLDS16(p.w + (a + i * wstep + (unsigned)stage * 64u), smem + D_WBASE + slot * D_WB + (wave * 4 + i) * 1024);
    };

    // ---- fragment read addresses ----
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned x_a = lds0 + j * 64 + ((g ^ ((-(j >> 2)) & 3)) << 4);                 // + mi*1024
    unsigned w_a[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int wrow = wave * 64 + (j >> 2) * 8 + par * 4 + (j & 3);                   // + (ni>>1)*32 rows
        w_a[par] = lds0 + D_WBASE + wrow * 64 + ((g ^ ((-(wrow >> 2)) & 3)) << 4);
    }

    floatx4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
    vec8 wa[4], wb[4], xw[4];
    const int S = p.stages;

    // hand-issued fragment reads with counted waits (see conv_mfma256.hip: hipcc would wait lgkmcnt(0) at
    // every use while an LDS-DMA is in flight)
#define VKD_DSR(dst, addr, OFF)                                                               \
    do {                                                                                      \
        if constexpr ((DBG & 4) != 0) asm volatile("" : "=v"(dst) : "v"(addr));               \
        else asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr));        \
    } while (0)
#define VKD_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg))
#define VKD_SB() __builtin_amdgcn_sched_barrier(0)
#define VKD_MMA_ROW(MI, XR, WF)                                                                      \
    do {                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if constexpr ((DBG & 8) == 0)                                                                \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[MI][ni] =                               \
            DuoT<ET>::mfma(WF[ni], XR, acc[MI][ni]);                                              \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)
#define VKD_READ_W(WF, so)                  \
    VKD_DSR(WF[0], w_a[0] + so, 0);         \
    VKD_DSR(WF[1], w_a[1] + so, 0);         \
    VKD_DSR(WF[2], w_a[0] + so, 2048);      \
    VKD_DSR(WF[3], w_a[1] + so, 2048)

    // The K loop is cut into PRE(s) = rows 0-4 of stage s, ending at the in-stage barrier, and POST(s) = rows 5-7
    // of stage s together with the first fragment reads of stage s+1.  A loop iteration is POST(s) + PRE(s+1), so
    // every loop boundary / branch sits right after an `s_waitcnt lgkmcnt(0)`: NO hand-issued ds_read is in flight
    // where hipcc may insert register copies (it does, at loop exits and back-edges; a copy of a register whose
    // ds_read has not landed moves stale data: tools/check_asm_hazards.py scans the ISA for exactly that).
    // FULL: steady state (constant waits, no branches); !FULL: first and last stages.
    // slot = s % 3 (uniform, carried by the caller); slot of s+2 = slot of s-1, slot of s+3 = slot of s.
    auto pre = [&](auto full_c, int s, int slot, const vec8 (&wcur)[4]) {
        constexpr bool FULL = decltype(full_c)::value;
        const int slot_p = slot == 0 ? D_NSLOT - 1 : slot - 1;       // weight slot of stage s+2 (= s-1)
        const unsigned xs = x_a + (unsigned)(XS == 4 ? (s & 3) : s % XS) * D_XB;
        const bool more = FULL || (s + 1 < S);
        const bool rw = FULL || (s + 2 < S);
        VKD_DSR(xw[3], xs, 3072); VKD_WAIT3(xw[0]); VKD_SB(); VKD_MMA_ROW(0, xw[0], wcur); VKD_SB();
        if (rw) req_w(s + 2, slot_p, 1);
        VKD_SB();
        VKD_DSR(xw[0], xs, 4096); VKD_WAIT3(xw[1]); VKD_SB(); VKD_MMA_ROW(1, xw[1], wcur); VKD_SB();
        if (rw) req_w(s + 2, slot_p, 2);
        VKD_SB();
        VKD_DSR(xw[1], xs, 5120); VKD_WAIT3(xw[2]); VKD_SB(); VKD_MMA_ROW(2, xw[2], wcur); VKD_SB();
        if (rw) req_w(s + 2, slot_p, 3);
        VKD_SB();
        VKD_DSR(xw[2], xs, 6144); VKD_WAIT3(xw[3]); VKD_SB(); VKD_MMA_ROW(3, xw[3], wcur); VKD_SB();
        VKD_DSR(xw[3], xs, 7168); VKD_WAIT3(xw[0]); VKD_SB(); VKD_MMA_ROW(4, xw[0], wcur); VKD_SB();
        // every read of stage s is issued; wait for them in straight-line code (a branch here would let hipcc
        // set up the tied operands with copies of registers whose reads are still in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
        VKD_SB();
        if (more) {
            // vmcnt: stage s+1 landed <=> only what was issued after its last piece may be outstanding: the two
            // pixel pieces of stage s-1+XS and the four weight pieces of stage s+2; barrier: the slots of stage s
            // are free
            if constexpr (STAMP) {     // where the stage's slack goes: DMA wait vs barrier wait (core-clock cycles)
                unsigned long ta, tb, tc;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
                if (FULL || s - 1 + XS < S)
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (s + 2 < S)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
                asm volatile("s_barrier" ::: "memory");
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc)::"memory");
                cyc_vm += tb - ta;
                cyc_bar += tc - tb;
            } else if (FULL || s - 1 + XS < S)
                asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else if (s + 2 < S)
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        VKD_SB();
    };
    // stage s+1 exists and has landed (PRE(s) waited): its weight fragments and first three pixel-row fragments
    // are read under rows 5-7 of stage s; stage s+3's first pieces go into the slot PRE(s)'s barrier freed.
    auto post = [&](auto full_c, int s, int slot, const vec8 (&wcur)[4], vec8 (&wnext)[4]) {
        constexpr bool FULL = decltype(full_c)::value;
        const int slot_n = slot == D_NSLOT - 1 ? 0 : slot + 1;       // weight slot of stage s+1
        const unsigned xn = x_a + (unsigned)(XS == 4 ? ((s + 1) & 3) : (s + 1) % XS) * D_XB, sn = (unsigned)slot_n * D_WB;
        const bool rx = FULL || (s + XS < S);
        const bool rw0 = FULL || (s + 3 < S);
        VKD_READ_W(wnext, sn);
        VKD_DSR(xw[0], xn, 0);
        VKD_SB();
        VKD_MMA_ROW(5, xw[1], wcur);
        VKD_SB();
        if (rx) req_x(s + XS, 0);
        VKD_DSR(xw[1], xn, 1024);
        VKD_SB();
        VKD_MMA_ROW(6, xw[2], wcur);
        VKD_SB();
        if (rx) req_x(s + XS, 1);
        VKD_DSR(xw[2], xn, 2048);
        VKD_SB();
        VKD_MMA_ROW(7, xw[3], wcur);
        VKD_SB();
        if (rw0) req_w(s + 3, slot, 0);
        VKD_SB();
    };
    auto last_rows = [&](const vec8 (&wcur)[4]) {
        VKD_MMA_ROW(5, xw[1], wcur);
        VKD_MMA_ROW(6, xw[2], wcur);
        VKD_MMA_ROW(7, xw[3], wcur);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;

    // ---- prologue: pixel rows of stages 0..XS-1, weights of stages 0 and 1 and the first weight piece of stage 2;
    // all of it is waited for (the pieces are in flight together, so this costs one latency) ----
#pragma unroll
    for (int st = 0; st < XS; ++st) {
        if (st < S) {
            req_x(st, 0);
            req_x(st, 1);
        }
    }
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        if (st < S) {
            req_w(st, st, 0);
            if (st < 2) {
                req_w(st, st, 1);
                req_w(st, st, 2);
                req_w(st, st, 3);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if constexpr (STAMP) {
        ts[1] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc0)::"memory");
    }
    VKD_READ_W(wa, 0u);
    VKD_DSR(xw[0], x_a, 0);
    VKD_DSR(xw[1], x_a, 1024);
    VKD_DSR(xw[2], x_a, 2048);
    auto next_slot = [](int sl) { return sl == D_NSLOT - 1 ? 0 : sl + 1; };
    pre(F_{}, 0, 0, wa);
    int s = 0, slot = 0;
    for (; s + XS + 1 < S && s + 5 < S; s += 2) {
        post(T_{}, s, slot, wa, wb);
        slot = next_slot(slot);
        pre(T_{}, s + 1, slot, wb);
        post(T_{}, s + 1, slot, wb, wa);
        slot = next_slot(slot);
        pre(T_{}, s + 2, slot, wa);
    }
    // S is even (the launcher checks): stages go in pairs, so the roles of the two weight-fragment sets are
    // static in every copy of the code and hipcc has no reason to shuffle them
    for (; s + 3 < S; s += 2) {
        post(F_{}, s, slot, wa, wb);
        slot = next_slot(slot);
        pre(F_{}, s + 1, slot, wb);
        post(F_{}, s + 1, slot, wb, wa);
        slot = next_slot(slot);
        pre(F_{}, s + 2, slot, wa);
    }
    post(F_{}, s, slot, wa, wb);
    slot = next_slot(slot);
    pre(F_{}, s + 1, slot, wb);
    last_rows(wb);
#undef VKD_DSR
#undef VKD_WAIT3
#undef VKD_MMA_ROW
#undef VKD_READ_W
#undef VKD_SB

    // ---- epilogue: + bias (+ residual) (ReLU) -> f16 through LDS, two halves of 64 rows x 256 channels ----
    asm volatile("s_barrier" ::: "memory");          // every wave has finished reading its fragments
    if constexpr (STAMP) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc1)::"memory");
        ts[2] = __builtin_amdgcn_s_memrealtime();
    }
    floatx4 *stg = reinterpret_cast<floatx4 *>(smem);
    auto load_res = [&](int h, vec8 (&rr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int it = tid + 256 * i, row = it >> 5, k8 = it & 31;
            const int m = min(m0 + h * 64 + row, p.M - 1);
            if (p.res)
                rr[i] = *reinterpret_cast<const vec8 *>(p.res + ((long)m * p.ldy + n0 + k8 * 8) * 2);
            else
                rr[i] = vec8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto stage_half = [&](auto h_c) {
        constexpr int h = decltype(h_c)::value;
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
            const int col = wave * 64 + qn * 32 + g * 8;              // tile-local channel of this lane's 8 values
            const floatx4 b0 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[0];
            const floatx4 b1 = reinterpret_cast<const floatx4 *>(p.bias + n0 + col)[1];
#pragma unroll
            for (int mq = 0; mq < 4; ++mq) {
                const int row = mq * 16 + j;
                const int c16 = col >> 2;
                stg[row * 64 + (c16 ^ (row & 7))] = acc[h * 4 + mq][2 * qn] + b0;
                stg[row * 64 + ((c16 + 1) ^ (row & 7))] = acc[h * 4 + mq][2 * qn + 1] + b1;
            }
        }
    };
    // fused spatial mean: a tile (128 rows) touches at most two images when HoWo >= 128.  This thread's 16 rows
    // of channel group k8 are summed per image from the f16 values a separate mean kernel would read back,
    // EXACTLY: v = floor(v) + frac, both parts accumulated as integers (|floor| <= 65504, frac * 2^24 < 2^24 and
    // exact because v has 11 significant bits), so the sum does not depend on where the tile boundaries fall
    // and an image's features do not depend on what else is in the batch.  A non-finite value poisons its
    // channel (sentinel -> NaN in pool_finish_kernel).
    int ph[2][8];
    unsigned pl[2][8];
    unsigned badmask = 0;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ph[sg][e] = 0;
            pl[sg][e] = 0u;
        }
    const int img0 = m0 / p.HoWo;
    const int m_split = (img0 + 1) * p.HoWo;         // first row of the tile's second image
    auto write_half = [&](int h, const vec8 (&rr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int it = tid + 256 * i, row = it >> 5, k8 = it & 31;
            const int m = m0 + h * 64 + row;
            const floatx4 v0 = stg[row * 64 + ((2 * k8) ^ (row & 7))];
            const floatx4 v1 = stg[row * 64 + ((2 * k8 + 1) ^ (row & 7))];
            vec8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = v0[e] + (float)rr[i][e], b = v1[e] + (float)rr[i][4 + e];
                if constexpr (GELU) {              // erf form (transformers ACT2FN["gelu"])
                    a = 0.5f * a * (1.0f + erff(a * 0.70710678118654752f));
                    b = 0.5f * b * (1.0f + erff(b * 0.70710678118654752f));
                } else if (p.relu) {
                    a = a > 0.f ? a : 0.f;
                    b = b > 0.f ? b : 0.f;
                }
                o[e] = (ET)a;
                o[4 + e] = (ET)b;
            }
            if (p.pool_part) {
                if (m < p.M) {
                    const bool second = m >= m_split;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (float)o[e];
                        const float fl = __builtin_floorf(v);
                        const int hi = (int)fl;
                        const unsigned lo = (unsigned)((v - fl) * 16777216.0f);
                        ph[0][e] += second ? 0 : hi;
                        ph[1][e] += second ? hi : 0;
                        pl[0][e] += second ? 0u : lo;
                        pl[1][e] += second ? lo : 0u;
                        badmask |= (__builtin_fabsf(v) <= 65504.0f ? 0u : 1u) << ((second ? 8 : 0) + e);
                    }
                }
            } else if (m < p.M) {
                *reinterpret_cast<vec8 *>(p.y + ((long)m * p.ldy + n0 + k8 * 8) * 2) = o;
            }
        }
    };
#define VKD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    vec8 r0[8], r1[8];
    load_res(0, r0);
    stage_half(std::integral_constant<int, 0>{});
    VKD_LDS_BARRIER();
    load_res(1, r1);
    write_half(0, r0);
    VKD_LDS_BARRIER();
    if constexpr (STAMP) ts[3] = __builtin_amdgcn_s_memrealtime();
    stage_half(std::integral_constant<int, 1>{});
    VKD_LDS_BARRIER();
    write_half(1, r1);
    if (p.pool_part) {
        // reduce the 8 row groups (tid >> 5) through LDS, then one thread per channel stores (hi, lo)
        VKD_LDS_BARRIER();                            // staging reads done
        int *red = reinterpret_cast<int *>(smem);     // [8 groups][2 images][256 channels][hi, lo]
        const int grp = tid >> 5, k8 = tid & 31;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool bad = (badmask >> (sg * 8 + e)) & 1u;
                red[(((grp * 2 + sg) * 256) + k8 * 8 + e) * 2] = bad ? 0x7fffffff : ph[sg][e];
                red[(((grp * 2 + sg) * 256) + k8 * 8 + e) * 2 + 1] = (int)pl[sg][e];
            }
        VKD_LDS_BARRIER();
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            int hi = 0;
            unsigned lo = 0u;
            bool bad = false;
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
                const int hv = red[(((gi * 2 + sg) * 256) + tid) * 2];
                bad |= hv == 0x7fffffff;
                hi += hv;
                lo += (unsigned)red[(((gi * 2 + sg) * 256) + tid) * 2 + 1];
            }
            int *dst = p.pool_part + (((long)m_tile * 2 + sg) * (p.n_tiles * D_BN) + n0 + tid) * 2;
            dst[0] = bad ? 0x7fffffff : hi;
            dst[1] = (int)lo;
        }
    }
#undef VKD_LDS_BARRIER
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[4] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long *o = p.stamps + (long)bid * 12;
            for (int i = 0; i < 5; ++i) o[i] = ts[i];
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
            o[7] = t;
            o[8] = cyc1 - cyc0;                                    // MFMA loop, core-clock cycles
            o[9] = cyc_vm;                                         // of which: waiting for the next stage's DMA
            o[10] = cyc_bar;                                       // of which: waiting for the other waves
            o[11] = p.stages;
        }
    }
}

bool conv_duo_dual_ok(const ConvArgs &a) {
    return a.x2 && !a.stem && (a.dt == VK_F16 || a.dt == VK_BF16) && a.out_dt == a.dt && a.kh == 1 && a.kw == 1 && a.pad == 0 && a.stride == 1 &&
           a.groups <= 1 && a.Cout % D_BN == 0 && a.ldy == a.Cout && a.Cin % 32 == 0 && a.Cin2 % 32 == 0 && a.Cin >= 32 &&
           (a.Cin + a.Cin2) % 64 == 0 &&
           a.Cin2 >= 32 && (long)a.N * a.H * a.W * std::max(a.Cin, a.Cin2) * 2 < (1L << 32);
}

bool conv_duo_pool_ok(const ConvArgs &a) {
    return a.pool_part && !a.stem && a.dt == VK_F16 && a.out_dt == VK_F16 && a.kh == 1 && a.kw == 1 && a.pad == 0 &&
           a.stride == 1 && a.groups <= 1 && a.Cout % D_BN == 0 && (a.Cin + (a.x2 ? a.Cin2 : 0)) % 64 == 0 && a.Cin >= 32 &&
           a.Ho * a.Wo >= D_BM &&
           a.Ho * a.Wo <= 255 &&                        // the 2^24-scaled fraction sums stay below 2^32
           (long)a.N * a.H * a.W * a.Cin * 2 < (1L << 32);
}

size_t conv_duo_pool_part_bytes(long M, int Cout) { return (size_t)((M + D_BM - 1) / D_BM) * 2 * Cout * 2 * sizeof(int); }

// out[n][c] = (exact sum of the tile partials that hold rows of image n) / HoWo, rounded once
__global__ void pool_finish_kernel(const int *__restrict__ part, int HoWo, int Cout, float *__restrict__ out) {
    const int n = blockIdx.x;
    const long r0 = (long)n * HoWo, r1 = r0 + HoWo - 1;
    const int t0 = (int)(r0 / D_BM), t1 = (int)(r1 / D_BM);
    for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
        long hi = 0;
        unsigned long lo = 0;
        bool bad = false;
        for (int t = t0; t <= t1; ++t) {
            const int sg = n - (int)((long)t * D_BM / HoWo);      // 0: the tile's first image, 1: its second
            const int *q = part + (((long)t * 2 + sg) * Cout + c) * 2;
            bad |= q[0] == 0x7fffffff;
            hi += q[0];
            lo += (unsigned)q[1];
        }
        const double sum = (double)hi + (double)lo * (1.0 / 16777216.0);
        out[(long)n * Cout + c] = bad ? __builtin_nanf("") : (float)(sum / (double)HoWo);
    }
}

// out[n][c] = (exact fp64 sum of image n's rows, conv_ws's fused-mean form) / HoWo, rounded once; a non-finite sum -- some output
// was not finite -- reads NaN, as above
__global__ void pool_finish64_kernel(const double *__restrict__ sums, int HoWo, int Cout, float *__restrict__ out) {
    const long n = blockIdx.x;
    for (int c = threadIdx.x; c < Cout; c += blockDim.x) {
        const double sum = sums[n * Cout + c];
        out[n * Cout + c] = __builtin_isfinite(sum) ? (float)(sum / (double)HoWo) : __builtin_nanf("");
    }
}

int launch_pool_finish(const float *part, int N, int HoWo, int Cin, int Cout, bool dual, float *out, hipStream_t stream) {
    if (conv_pool_sums_f64(N, HoWo, Cin, Cout, dual))
        hipLaunchKernelGGL(pool_finish64_kernel, dim3(N), dim3(256), 0, stream, (const double *)part, HoWo, Cout, out);
    else
        hipLaunchKernelGGL(pool_finish_kernel, dim3(N), dim3(256), 0, stream, (const int *)part, HoWo, Cout, out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

bool conv_duo_eligible(const ConvArgs &a) {
    if (a.pool_part) return conv_duo_pool_ok(a);
    if (a.x2) return conv_duo_dual_ok(a);
    const char *v = getenv("VK_CONV_DUO");               // "0" disables (A/B switch, re-read per call)
    if (v && v[0] == '0') return false;
    if (a.stem || (a.dt != VK_F16 && a.dt != VK_BF16) || a.out_dt != a.dt) return false;
    if (a.relu > 2 || (a.relu == 2 && a.dt != VK_BF16)) return false;     // GELU epilogue: bf16 build only; no tanh
    if (a.kh != 1 || a.kw != 1 || a.pad != 0) return false;
    if (a.Cout % D_BN != 0 || a.ldy != a.Cout || a.Cin % 64 != 0) return false;
    if ((long)a.N * a.H * a.W * a.Cin * 2 >= (1L << 32)) return false;   // 32-bit DMA offsets
    const long M = (long)a.N * a.Ho * a.Wo;
    if (M < 8 * D_BM) return false;
    // measured (interleaved A/B, one device): -10...-19 % on every K <= 512 layer and on the small-grid K = 1024
    // layers; the large-grid K >= 1024 layers (Res5 conv1 / shortcut) are MFMA-loop bound and equal or 1-2 % slower
    if (v && v[0] == '1') return true;                   // "1" forces it wherever it is legal (A/B, tests)
    return a.Cin <= 512 || M < 400000;
}

int launch_conv_duo(const ConvArgs &a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<_Float16, false, 3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<__bf16, false, 3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<_Float16, false, 3, 0, false, 1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<__bf16, false, 3, 0, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<_Float16, true, 3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)));
#define VKD_DBG_ATTR(D) VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_duo_kernel<_Float16, true, 3, D>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, d_smem(3)))
        VKD_DBG_ATTR(1); VKD_DBG_ATTR(3); VKD_DBG_ATTR(4); VKD_DBG_ATTR(7); VKD_DBG_ATTR(8); VKD_DBG_ATTR(11); VKD_DBG_ATTR(12);
#undef VKD_DBG_ATTR
        attr_set = true;
    }
    DuoK k;
    k.x = (const char *)a.x;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    k.H = a.H;
    k.W = a.W;
    k.Ho = a.Ho;
    k.Wo = a.Wo;
    k.HoWo = a.Ho * a.Wo;
    const long M = (long)a.N * a.Ho * a.Wo;
    VK_REQUIRE(M > 0 && M < (1L << 31) - D_BM, VK_EINVAL, "conv_duo: M=%ld out of range", M);
    VK_REQUIRE((long)a.N * a.H * a.W * a.Cin * 2 < (1L << 32) && (long)a.Cout * (a.Cin + a.Cin2) * 2 < (1L << 32), VK_EINVAL,
               "conv_duo: tensor beyond the 32-bit DMA offsets");
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.ldy = a.ldy;
    k.stride = a.stride;
    k.x2 = (const char *)a.x2;
    k.cin2_bytes = a.x2 ? a.Cin2 * 2 : 0;
    k.split = a.Cin / 32;
    k.stages = (a.Cin + (a.x2 ? a.Cin2 : 0)) / 32;
    VK_REQUIRE(k.stages >= 2 && k.stages % 2 == 0, VK_EINVAL, "conv_duo: K = %d is not a whole number of 64-channel pairs of stages",
               k.stages * 32);
    k.wrow_bytes = (a.Cin + (a.x2 ? a.Cin2 : 0)) * 2;
    k.relu = a.relu;
    k.m_tiles = ceil_div(k.M, D_BM);
    k.n_tiles = a.Cout / D_BN;
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    const dim3 grid(k.m_tiles * k.n_tiles), block(256);
    k.pool_part = (int *)a.pool_part;
    k.stamps = nullptr;
#ifdef VK_ABLATION      // stamp / timing-only builds: tools/ builds only (make ABLATION=1)
    if (const char *sf = getenv("VK_DUO_STAMPS")) {      // diagnostic: one launch, phase stamps appended to the file
        const size_t nb = (size_t)grid.x * 12 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        const int dbg = getenv("VK_DUO_DBG") ? atoi(getenv("VK_DUO_DBG")) : 0;
        if (false) {}
#define VKD_DBG_RUN(D) else if (dbg == D) hipLaunchKernelGGL((conv_duo_kernel<_Float16, true, 3, D>), grid, block, d_smem(3), stream, k)
        VKD_DBG_RUN(1); VKD_DBG_RUN(3); VKD_DBG_RUN(4); VKD_DBG_RUN(7); VKD_DBG_RUN(8); VKD_DBG_RUN(11); VKD_DBG_RUN(12);
#undef VKD_DBG_RUN
        else
            hipLaunchKernelGGL((conv_duo_kernel<_Float16, true, 3>), grid, block, d_smem(3), stream, k);
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> h((size_t)grid.x * 12);
        VK_CHECK_HIP(hipMemcpy(h.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# launch M=%d cout=%d cin=%d grid=%u\n", k.M, a.Cout, a.Cin, grid.x);
            for (unsigned b = 0; b < grid.x; ++b) {
                fprintf(f, "%u", b);
                for (int i = 0; i < 12; ++i) fprintf(f, " %lu", h[(size_t)b * 12 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    } else
#endif
        if (a.dt == VK_BF16 && a.relu == 2)
            hipLaunchKernelGGL((conv_duo_kernel<__bf16, false, 3, 0, true>), grid, block, d_smem(3), stream, k);
        else if (a.dt == VK_BF16)
            hipLaunchKernelGGL((conv_duo_kernel<__bf16, false, 3>), grid, block, d_smem(3), stream, k);
        else if (a.concurrent)
            hipLaunchKernelGGL((conv_duo_kernel<_Float16, false, 3, 0, false, 1>), grid, block, d_smem(3), stream, k);
        else
            hipLaunchKernelGGL((conv_duo_kernel<_Float16, false, 3>), grid, block, d_smem(3), stream, k);
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const int K = a.Cin + (a.x2 ? a.Cin2 : 0);
        tm->recs.push_back({a.concurrent ? 6 : 5, 2.0 * (double)k.M * a.Cout * K, e0, e1, k.M, a.Cout, K, 1, a.stride,
                            2.0 * ((double)a.N * a.H * a.W * K + (double)k.M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * K)});
    }
    return VK_OK;
}

}  // namespace vk
