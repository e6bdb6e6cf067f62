// 256x256-tile implicit-GEMM convolution for the FLOP-heavy layers (Res5 head, res3/res4, RPN 3x3):
// same contract as conv_mfma.hip (NHWC fp16 in, fp32 accumulate, + bias (+ residual)(+ ReLU),
// fp16 out), built for one 512-thread workgroup (8 waves, 2 x 4) per CU.
//
// Structure (CDNA4-specific):
//   * K advances in stages of 32 channels (64 B per row).  LDS is a RING of 4 stage slots
//     (4 x (256 pixel rows + 256 channel rows) x 64 B = 128 KiB); stage s+4 is requested while
//     stage s is finishing, so three stages of HBM/L2 latency are always in flight per CU.
//   * global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4, 16 B per lane, no
//     VGPR staging).  The LDS image must be lane-linear per wave-instruction, so the XOR swizzle
//     that makes the ds_read_b128 fragment reads conflict-free (64-B rows: phys chunk =
//     chunk ^ (-(row>>2) & 3)) is applied to each lane's SOURCE address.  im2col gather = the
//     per-lane source address; out-of-image taps and rows beyond M read a zero page.
//   * one raw s_barrier per stage (32 MFMAs per wave), placed three quarters through the stage:
//     wait vmcnt(8) (stage s+1 landed; s+2, s+3 still in flight) -> lgkmcnt(0) -> barrier -> request
//     stage s+4 into the slot of stage s (every wave has finished reading it) -> read the weight
//     fragments of stage s+1.  Pixel-row fragments stream through a 4-deep register window, read
//     two row-tiles ahead of the 4 MFMAs (v_mfma_f32_16x16x32_f16) that consume them, across the
//     stage boundary, so LDS latency is always covered by MFMAs of the same wave.
//   * per-wave output 128 x 64 (8 x 4 accumulator tiles, 128 accumulator registers); the weight
//     fragment is the MFMA A operand so a lane owns 8 consecutive output channels = one 16-byte
//     NHWC store, as in conv_mfma.hip.
#include <type_traits>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct Conv256K {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    const char *zero;    // >= 16 B of zeros (source of padded taps / rows beyond M)
    const char *x2;      // DUAL builds: second K segment of a 1x1 conv (stages >= st_per_tap): rows of [M, cin2]
    int cin2_bytes;
    int H, W, Ho, Wo, HoWo, M;
    int cin_bytes, ldy;
    int kw, stride, pad, dil;
    int stages, st_per_tap;   // K stages of 32 channels
    int ntaps;                // kh * kw
    int wrow_bytes;
    int relu;
    int m_tiles, n_tiles;
};

constexpr int R_BM = 256, R_BN = 256, R_ROWB = 64, R_NSLOT = 4;
constexpr int R_XB = R_BM * R_ROWB;          // 16 KiB
constexpr int R_SLOT = 2 * R_XB;             // 32 KiB
constexpr int R_SMEM = R_NSLOT * R_SLOT;     // 128 KiB

#define VK_GLDS16(gptr, lptr)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// DBG != 0 are diagnostic / A-B builds selected with VK_CONV256_DBG (never used by the product path):
// 1 = no LDS-DMA in the steady state, 2 = no pixel-row fragment reads, 3 = both (timing only, WRONG results);
// 16 = taps innermost (no gain), 32 / 64 = weight / pixel DMA pieces all read ONE cached line (timing only),
// 4 = static s_setprio(1) for waves 4-7 (no effect measured), 8 = WITHOUT the s_setprio pair around each
// 4-MFMA group (the pair is worth +1.3 % median, interleaved A/B in one process on one device).
// DUAL: 1x1, stride 1, two inputs: K = Cin | Cin2 (conv3 + projection shortcut as one GEMM, as in conv_mfma_duo.hip -- same
// K order and epilogue, so the bits are the same; at K = 512 + 1024 the 256 x 256 tile needs a third less operand fill per
// MFMA than the 128 x 256 one)
template <int DBG, bool DUAL = false>
__global__ __launch_bounds__(512, 2) void conv_mfma256_kernel(Conv256K p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware (bijective) workgroup -> tile map (see conv_mfma.hip).  A persistent-workgroup variant
    // (tile loop inside the kernel) was tried and rejected: the loop-carried state pushed the kernel over
    // the 256-VGPR budget (79 spilled registers, 8-25 % slower on every shape).
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tile = t % p.n_tiles, m_tile = t / p.n_tiles;
    const int m0 = m_tile * R_BM, n0 = n_tile * R_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, j = lane & 15;

    // ---- LDS-DMA source state: this lane feeds rows (wave*2+i)*16 + (lane>>2), i = 0,1 ----
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);   // logical 16-B chunk whose bytes land at phys chunk lane&3
    long a_off[2], a_off2[2];
    int bh[2], bw[2];
    const char *wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lrow;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n_img = mm / p.HoWo;
        const int rem = mm - n_img * p.HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        a_off[i] = ((long)(n_img * p.H + h0) * p.W + w0) * p.cin_bytes + lchunk * 16;
        a_off2[i] = DUAL ? (long)mm * p.cin2_bytes + lchunk * 16 : 0;
        bh[i] = ok ? h0 : -(1 << 28);
        bw[i] = w0;
        wsrc[i] = p.w + (long)(n0 + row) * p.wrow_bytes + lchunk * 16;
    }
    const int dma_x0 = (wave * 2) * 1024;               // byte offset of this wave's first X row block in a slot
    const int dma_w0 = R_XB + (wave * 2) * 1024;

    // An LDS-DMA instruction is slow to ISSUE (~60-180 cycles), so the 4 pieces of a stage are never
    // issued back to back: each one follows a group of 4 MFMAs that keeps the matrix pipe busy meanwhile.
    int khi = 0, kwi = 0, kc = 0;   // tap / channel-stage of the NEXT pixel-row request
    const char *xsrc[2];
    auto prep_x = [&]() {           // source addresses of the next stage's two pixel-row pieces
        if constexpr (DUAL) {       // kc counts the 32-channel stages of both inputs
            const bool second = kc >= p.st_per_tap;
            const char *base = second ? p.x2 : p.x;
            const long toff = (long)(second ? kc - p.st_per_tap : kc) * R_ROWB;
#pragma unroll
            for (int i = 0; i < 2; ++i) xsrc[i] = bh[i] >= 0 ? base + (second ? a_off2[i] : a_off[i]) + toff : p.zero;
            kc += 1;
            return;
        }
        const int dh = khi * p.dil, dw = kwi * p.dil;
        const long toff = ((long)dh * p.W + dw) * p.cin_bytes + kc * R_ROWB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = (unsigned)(bh[i] + dh) < (unsigned)p.H && (unsigned)(bw[i] + dw) < (unsigned)p.W;
            xsrc[i] = (ok && !(DBG & 64)) ? p.x + a_off[i] + toff : p.zero;   // DBG 64: every pixel piece reads one cached line
        }
        // branch-free advance of (kernel row, kernel col, channel stage): keeps the K loop one basic block
        if constexpr (DBG & 16) {   // experiment: taps innermost (consecutive stages re-read almost the same pixel lines)
            kwi += 1;
            const int c1 = (kwi == p.kw) ? 1 : 0;
            kwi *= (1 - c1);
            khi += c1;
            const int c2 = (khi * p.kw == p.ntaps) ? 1 : 0;
            khi *= (1 - c2);
            kc += c2;
        } else {
            kc += 1;
            const int c1 = (kc == p.st_per_tap) ? 1 : 0;
            kc *= (1 - c1);
            kwi += c1;
            const int c2 = (kwi == p.kw) ? 1 : 0;
            kwi *= (1 - c2);
            khi += c2;
        }
    };
    int wtap = 0, wkc = 0;          // (tap, channel stage) of the NEXT weight request (DBG & 16 order)
    auto req_x = [&](int stage, int i) {
        if constexpr (DBG & 1) return;
        VK_GLDS16(xsrc[i], smem + (stage & (R_NSLOT - 1)) * R_SLOT + dma_x0 + i * 1024);
    };
    auto req_w = [&](int stage, int i) {
        if constexpr (DBG & 1) return;
        long woff = (long)stage * R_ROWB;
        if constexpr (DBG & 16) {
            woff = (long)(wtap * p.st_per_tap + wkc) * R_ROWB;
            if (i == 1) {
                wtap += 1;
                const int c = (wtap == p.ntaps) ? 1 : 0;
                wtap *= (1 - c);
                wkc += c;
            }
        }
        VK_GLDS16((DBG & 32) ? p.zero : wsrc[i] + woff, smem + (stage & (R_NSLOT - 1)) * R_SLOT + dma_w0 + i * 1024);
    };

    // ---- fragment read addresses (bytes inside a slot) ----
    const int sx = (-(j >> 2)) & 3;
    const int x_addr = (wr * 128 + j) * R_ROWB + ((g ^ sx) << 4);                       // + mi*1024
    int w_addr[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int wrow = wc * 64 + (j >> 2) * 8 + par * 4 + (j & 3);                    // + (ni>>1)*32 rows
        w_addr[par] = R_XB + wrow * R_ROWB + ((g ^ ((-(wrow >> 2)) & 3)) << 4);
    }

    if constexpr (DBG & 4) {   // experiment: static priority for the younger half (guide T5 static form)
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    floatx4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Fragment registers: two weight sets (current / next stage) and a 4-deep rotating window of
    // pixel-row fragments read three row-tiles ahead of the MFMAs that consume them.
    //
    // hipcc waits lgkmcnt(0) at every use of a compiler-issued ds_read while an LDS-DMA is in flight
    // (verified in the ISA), which would expose the full LDS latency every row.  The fragment reads
    // and their COUNTED waits are therefore issued by hand: LDS returns in order, so with reads issued
    // as  w'0..w'3, x0, x1, x2, [x3 | wait 3 | mma0], [x4 | wait 3 | mma1], ...  "all but the 3 newest"
    // is exactly "row r and everything before it".  Each wait names its row register "+v" so the
    // MFMAs that consume it cannot be scheduled above it.
    half8 wa[4], wb[4], xw[4];
    const int S = p.stages;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const unsigned x_a = lds0 + x_addr, w_a0 = lds0 + w_addr[0], w_a1 = lds0 + w_addr[1];

#define VK_DSR(dst, addr, OFF)                                                                \
    do {                                                                                      \
        if constexpr (DBG & 2)                                                                \
            asm volatile("" : "+v"(dst) : "v"(addr));                                         \
        else                                                                                  \
            asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr));         \
    } while (0)
#define VK_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg))
#define VK_MMA_ROW(MI, XR, WF)                                                                       \
    do {                                                                                             \
        if constexpr (!(DBG & 8)) __builtin_amdgcn_s_setprio(1);                                     \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[MI][ni] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[ni], XR, acc[MI][ni], 0, 0, 0);                \
        if constexpr (!(DBG & 8)) __builtin_amdgcn_s_setprio(0);                                     \
    } while (0)
#define VK_READ_W(WF, so)          \
    VK_DSR(WF[0], w_a0 + so, 0);   \
    VK_DSR(WF[1], w_a1 + so, 0);   \
    VK_DSR(WF[2], w_a0 + so, 2048); \
    VK_DSR(WF[3], w_a1 + so, 2048)

    // The K loop is cut into PRE(s) = rows 0-4 of stage s, ending at the in-stage barrier, and POST(s) = rows 5-7
    // of stage s together with the first fragment reads of stage s+1.  A loop iteration is POST(s) + PRE(s+1), so
    // every loop boundary / branch sits right after an `s_waitcnt lgkmcnt(0)`: NO hand-issued ds_read is in flight
    // where hipcc may insert register copies (it does, at loop exits and back-edges; a copy of a register whose
    // ds_read has not landed moves stale data: tools/check_asm_hazards.py scans the ISA for exactly that).
    // FULL: steady state (stage s+4 exists): constant waits, no branches.  !FULL: first and last stages.
    auto pre = [&](auto full_c, int s, const half8 (&wcur)[4]) {
        constexpr bool FULL = decltype(full_c)::value;
        const bool more = FULL || (s + 1 < S);
        const unsigned so = (unsigned)(s & (R_NSLOT - 1)) * R_SLOT;           // this stage's slot
        const unsigned xs = x_a + so;
        const bool rw = FULL || (s + 3 < S);      // weight pieces of stage s+3 ride on rows 0, 1
        const bool rx = FULL || (s + 4 < S);      // pixel pieces of stage s+4 ride on rows 5, 6 (after the barrier)
        // rows 0..4: read row r+3 of this stage, wait for row r (all but the 3 newest reads), 4 MFMAs.
        VK_DSR(xw[3], xs, 3072); VK_WAIT3(xw[0]); __builtin_amdgcn_sched_barrier(0); VK_MMA_ROW(0, xw[0], wcur); __builtin_amdgcn_sched_barrier(0);
        if (rw) req_w(s + 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        VK_DSR(xw[0], xs, 4096); VK_WAIT3(xw[1]); __builtin_amdgcn_sched_barrier(0); VK_MMA_ROW(1, xw[1], wcur); __builtin_amdgcn_sched_barrier(0);
        if (rw) req_w(s + 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        VK_DSR(xw[1], xs, 5120); VK_WAIT3(xw[2]); __builtin_amdgcn_sched_barrier(0); VK_MMA_ROW(2, xw[2], wcur); __builtin_amdgcn_sched_barrier(0);
        if (rx) prep_x();                        // address arithmetic for the rows-5/6 pieces, off the critical path
        VK_DSR(xw[2], xs, 6144); VK_WAIT3(xw[3]); __builtin_amdgcn_sched_barrier(0); VK_MMA_ROW(3, xw[3], wcur); __builtin_amdgcn_sched_barrier(0);
        VK_DSR(xw[3], xs, 7168); VK_WAIT3(xw[0]); __builtin_amdgcn_sched_barrier(0); VK_MMA_ROW(4, xw[0], wcur); __builtin_amdgcn_sched_barrier(0);
        // every read of stage s is issued; wait for them in straight-line code (a branch here would let hipcc
        // set up the tied operands with copies of registers whose reads are still in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            // every wave's DMA of stage s+1 has landed (vmcnt: all but the pieces of stages s+2, s+3: 4 each; the
            // pixel pieces of stage s+4 are only issued after this wait), and once every wave is here
            // slot (s+4)&3 == slot(s) is free
            if (FULL || s + 3 < S)
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else if (s + 2 < S)
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // stage s+1 exists and has landed: its weight fragments and first three pixel-row fragments are read under
    // rows 5-7 of stage s; the pixel pieces of stage s+4 go into the slot PRE(s)'s barrier freed
    auto post = [&](auto full_c, int s, const half8 (&wcur)[4], half8 (&wnext)[4]) {
        constexpr bool FULL = decltype(full_c)::value;
        const unsigned sn = (unsigned)((s + 1) & (R_NSLOT - 1)) * R_SLOT;     // next stage's slot
        const unsigned xn = x_a + sn;
        const bool rx = FULL || (s + 4 < S);
        VK_READ_W(wnext, sn);
        VK_DSR(xw[0], xn, 0);
        __builtin_amdgcn_sched_barrier(0);
        VK_MMA_ROW(5, xw[1], wcur);
        __builtin_amdgcn_sched_barrier(0);
        if (rx) req_x(s + 4, 0);
        VK_DSR(xw[1], xn, 1024);
        __builtin_amdgcn_sched_barrier(0);
        VK_MMA_ROW(6, xw[2], wcur);
        __builtin_amdgcn_sched_barrier(0);
        if (rx) req_x(s + 4, 1);
        VK_DSR(xw[2], xn, 2048);
        __builtin_amdgcn_sched_barrier(0);
        VK_MMA_ROW(7, xw[3], wcur);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto last_rows = [&](const half8 (&wcur)[4]) {
        VK_MMA_ROW(5, xw[1], wcur);
        VK_MMA_ROW(6, xw[2], wcur);
        VK_MMA_ROW(7, xw[3], wcur);
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;

    // prologue: stages 0..2 completely, and the pixel rows of stage 3 (its weight rows ride on stage 0)
    int issued = 0;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st < S) {
            prep_x();
            req_x(st, 0);
            req_x(st, 1);
            issued += 2;
            if (st < 3) {
                req_w(st, 0);
                req_w(st, 1);
                issued += 2;
            }
        }
    }
    // stage 0 (the 4 oldest pieces) must have landed
    if (issued == 14)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (issued == 12)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (issued == 8)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    VK_READ_W(wa, 0u);
    VK_DSR(xw[0], x_a, 0);
    VK_DSR(xw[1], x_a, 1024);
    VK_DSR(xw[2], x_a, 2048);
    pre(F_{}, 0, wa);
    int s = 0;
    for (; s + 6 < S; s += 2) {
        post(T_{}, s, wa, wb);
        pre(T_{}, s + 1, wb);
        post(T_{}, s + 1, wb, wa);
        pre(T_{}, s + 2, wa);
    }
    // S is even (the launcher checks): stages go in pairs, so the roles of the two weight-fragment sets are static
    for (; s + 3 < S; s += 2) {
        post(F_{}, s, wa, wb);
        pre(F_{}, s + 1, wb);
        post(F_{}, s + 1, wb, wa);
        pre(F_{}, s + 2, wa);
    }
    post(F_{}, s, wa, wb);
    pre(F_{}, s + 1, wb);
    last_rows(wb);
#undef VK_DSR
#undef VK_WAIT3
#undef VK_MMA_ROW
#undef VK_READ_W

    // ---- epilogue: + bias (+ residual) (ReLU) -> f16, through LDS so that HBM sees whole lines ----
    // In the accumulator layout a wave-instruction touches 16 rows x 64 B (half cache lines), and the
    // per-CU memory pipeline (not HBM) bounds the epilogue of the small-K layers.  The ring is dead now,
    // so the f32 tile goes through LDS in two 128-row halves ([128][256] f32 = 128 KiB, 16-B chunks
    // XOR-swizzled by row) and is read back row-contiguous: every residual load and every store of a
    // wave covers 2 rows x 512 B = 8 whole 128-B lines.
    asm volatile("s_barrier" ::: "memory");          // every wave has finished reading its fragments
    floatx4 *stg = reinterpret_cast<floatx4 *>(smem);
    // Residual loads are issued a phase ahead of their use (half 0 before the staging writes, half 1
    // before half 0 is written out) and the barriers are raw (lgkmcnt only), so the HBM latency of one
    // half hides behind the LDS work of the other instead of adding up.
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
                const int col = wc * 64 + qn * 32 + g * 8;              // tile-local channel of this lane's 8 values
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
#define VK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    half8 r0[8], r1[8];
    load_res(0, r0);
    stage_half(0);
    VK_LDS_BARRIER();
    load_res(1, r1);
    write_half(0, r0);
    VK_LDS_BARRIER();
    stage_half(1);
    VK_LDS_BARRIER();
    write_half(1, r1);
#undef VK_LDS_BARRIER
}

bool conv256_dual_ok(const ConvArgs &a) {
    const char *v = getenv("VK_CONV256_DUAL");           // "0": the two-per-CU kernel takes the dual-source form (A/B switch)
    if (v && v[0] == '0') return false;
    if (!a.x2 || a.stem || a.pool_part || a.groups > 1 || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return false;
    if (a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride != 1) return false;
    if (a.Cout % R_BN != 0 || a.ldy != a.Cout || a.Cin % 32 != 0 || a.Cin2 % 32 != 0 || (a.Cin + a.Cin2) % 64 != 0) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    // measured on the Res5 shape (512 | 1024 -> 2048): worth it when the K loop is long; short K stays two-per-CU
    return M >= 4 * R_BM && a.Cin + a.Cin2 >= 1024;
}

bool conv256_eligible(const ConvArgs &a) {
    static const bool disabled = getenv("VK_DISABLE_CONV256") != nullptr;
    if (disabled || a.stem) return false;
    if (a.relu > 1) return false;             // only the 128-tile kernel has the GELU / tanh epilogues
    if (a.dt != VK_F16 || a.out_dt != VK_F16) return false;
    if (a.Cout % R_BN != 0 || a.ldy != a.Cout) return false;
    if (a.Cin % 64 != 0) return false;        // an even number of 32-channel stages (the K loop runs them in pairs)
    const long M = (long)a.N * a.Ho * a.Wo;
    return M >= 4 * R_BM;
}

int launch_conv256(const ConvArgs &a, hipStream_t stream) {
    static char *zero_page = nullptr;
    if (!zero_page) {
        VK_CHECK_HIP(hipMalloc((void **)&zero_page, 256));
        VK_CHECK_HIP(hipMemset(zero_page, 0, 256));
    }
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<32>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<96>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
        attr_set = true;
    }
    Conv256K k;
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
    VK_REQUIRE(M > 0 && M < (1L << 31) - R_BM, VK_EINVAL, "conv256: M=%ld out of range", M);
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.x2 = (const char *)a.x2;
    k.cin2_bytes = a.x2 ? a.Cin2 * 2 : 0;
    k.ldy = a.ldy;
    k.kw = a.kw;
    k.stride = a.stride;
    k.pad = a.pad;
    k.dil = a.dil;
    k.st_per_tap = a.Cin / 32;
    k.stages = a.x2 ? (a.Cin + a.Cin2) / 32 : a.kh * a.kw * k.st_per_tap;
    k.ntaps = a.kh * a.kw;
    k.wrow_bytes = a.x2 ? (a.Cin + a.Cin2) * 2 : a.kh * a.kw * a.Cin * 2;
    k.relu = a.relu;
    k.m_tiles = ceil_div(k.M, R_BM);
    k.n_tiles = a.Cout / R_BN;
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    const int dbg = getenv("VK_CONV256_DBG") ? atoi(getenv("VK_CONV256_DBG")) : 0;   // re-read: lets one process A/B variants
    const dim3 grid(k.m_tiles * k.n_tiles), block(512);
    if (a.x2) {
        static bool dual_attr = false;
        if (!dual_attr) {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma256_kernel<0, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM));
            dual_attr = true;
        }
        hipLaunchKernelGGL((conv_mfma256_kernel<0, true>), grid, block, R_SMEM, stream, k);
    } else
#ifdef VK_ABLATION      // timing-only ablation builds (WRONG results): tools/ builds only (make ABLATION=1)
    switch (dbg) {
        case 1: hipLaunchKernelGGL(conv_mfma256_kernel<1>, grid, block, R_SMEM, stream, k); break;
        case 2: hipLaunchKernelGGL(conv_mfma256_kernel<2>, grid, block, R_SMEM, stream, k); break;
        case 3: hipLaunchKernelGGL(conv_mfma256_kernel<3>, grid, block, R_SMEM, stream, k); break;
        case 4: hipLaunchKernelGGL(conv_mfma256_kernel<4>, grid, block, R_SMEM, stream, k); break;
        case 8: hipLaunchKernelGGL(conv_mfma256_kernel<8>, grid, block, R_SMEM, stream, k); break;
        case 16: hipLaunchKernelGGL(conv_mfma256_kernel<16>, grid, block, R_SMEM, stream, k); break;
        case 32: hipLaunchKernelGGL(conv_mfma256_kernel<32>, grid, block, R_SMEM, stream, k); break;
        case 64: hipLaunchKernelGGL(conv_mfma256_kernel<64>, grid, block, R_SMEM, stream, k); break;
        case 96: hipLaunchKernelGGL(conv_mfma256_kernel<96>, grid, block, R_SMEM, stream, k); break;
        default: hipLaunchKernelGGL(conv_mfma256_kernel<0>, grid, block, R_SMEM, stream, k);
    }
#else
    {
        (void)dbg;
        hipLaunchKernelGGL(conv_mfma256_kernel<0>, grid, block, R_SMEM, stream, k);
    }
#endif
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const int K = a.kh * a.kw * a.Cin + (a.x2 ? a.Cin2 : 0);
        tm->recs.push_back({a.concurrent ? 6 : (a.x2 ? 9 : 0), 2.0 * (double)k.M * a.Cout * K, e0, e1, k.M, a.Cout, a.x2 ? K : a.Cin, a.kh * a.kw, a.stride,
                            2.0 * ((double)a.N * a.H * a.W * (a.Cin + (a.x2 ? a.Cin2 : 0)) + (double)k.M * a.Cout * (a.res ? 2 : 1) +
                                   (double)a.Cout * K)});
    }
    return VK_OK;
}

}  // namespace vk
