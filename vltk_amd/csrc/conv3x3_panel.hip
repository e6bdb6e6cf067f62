// 3x3 convolution (stride 1, pad == dilation) with an LDS-RESIDENT INPUT PANEL: the conv-native kernel
// for the layers where an im2col GEMM wastes the most bytes (Res5 conv2 with dilation 2, res4 conv2, RPN 3x3).
//
// Why: the 256x256 im2col ring kernel (conv_mfma256.hip) is bound by the supply of distinct cache lines
// through the CU's L1-miss path (~15 B/clk/CU) and an im2col GEMM fetches every input pixel
// NINE times (once per tap).  Here K is walked channel-stage-major / tap-minor: the 32-channel slab of the
// tile's pixels PLUS HALO (all pixels any tap can touch) is brought into LDS ONCE per channel stage and the
// nine taps read it at nine row shifts; only the weights stream per (tap, stage).  Bytes through the CU per
// 32-channel stage: 9 x 16 KiB weights + ~24-32 KiB panel  vs  9 x 32 KiB  (-40...-45 %).
//
//   * LDS: 2 panel buffers of PP*128 rows x 64 B (PP = 3 or 4: halo 64 / 128 rows >= dil*(W+1)) + a ring of
//     6 weight slots (256 channels x 64 B): 144 / 160 KiB.  Same 64-B-row XOR swizzle as the ring kernel,
//     applied on the DMA source address; the fragment address is recomputed per tap (the shift moves the key).
//   * a tap is a UNIFORM row shift ((kh-1)*W + (kw-1))*dil of the fragment reads; pixels whose tap falls outside
//     the image (or into the neighbouring image, or rows >= M) are zeroed in registers from a precomputed
//     72-bit per-lane validity mask (4 v_cndmask per fragment).
//   * per step (one tap of one stage) = 32 MFMAs per wave, one raw barrier, 2 weight DMA pieces per wave,
//     plus one panel piece of the NEXT channel stage on the first PP taps; hand-issued ds_read_b128 with
//     counted lgkmcnt, counted vmcnt (the count depends only on the tap index: compile-time).
//   * same per-wave geometry (2x4 waves, 128x64 per wave, 16x16x32 f16 MFMA) and the same LDS-staged
//     whole-row epilogue as conv_mfma256.hip.
#include <cstdio>
#include <type_traits>
#include <vector>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

struct PanelK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int H, W, HW, M;
    int cin_bytes, ldy;
    int dil;
    int cstages;              // Cin / 32 (even)
    int wrow_bytes;           // 9 * Cin * 2
    int relu;
    int m_tiles, n_tiles;
    unsigned long *stamps;    // DBG & 4 builds: 6 words per workgroup
    // Dynamic tail (round 3; two-panel launches of many rounds only).  The dispatcher gives every XCD the same number of workgroups
    // (blockIdx mod 8) and the XCDs of one chip differ by ~4.5 % in speed at the power limit (stamps: the last workgroup of the slowest
    // ends 300 us after the fastest's on a 7.1 ms launch, 2 % of the CU time idle).  Workgroups [0, static_tiles) take the tile of their
    // index as before; the grid carries `total - static_tiles` + spare further workgroups, each of which takes the next tail tile from an
    // atomic counter when it STARTS (so a fast XCD does more of them) and leaves at once when none is left.
    unsigned *tile_ctr;       // nullptr: every workgroup is its own tile
    int static_tiles;
    unsigned ctr_last;        // the launch's last fetch (one per tail workgroup): whoever draws it zeroes the word for its next user
};

constexpr int P_NW = 6;                     // weight ring slots
constexpr int P_WSLOT = 256 * 64;           // 16 KiB

#define VKP_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// panel pieces issued in the window of P_NW-2 steps before tap J (taps of the previous stage when negative)
template <int J, int PP>
constexpr int panel_in_window() {
    int c = 0;
    for (int d = 1; d <= P_NW - 2; ++d) {
        int t = ((J - d) % 9 + 9) % 9;
        if (t < PP) ++c;
    }
    return c;
}
// weight pieces (x2) issued in that window during the LAST channel stage (its taps >= 3 issue nothing)
template <int J>
constexpr int last_w_in_window() {
    int c = 0;
    for (int d = 1; d <= P_NW - 2; ++d) {
        int t = J - d;                     // < 0: previous stage, always issues
        if (t < 0 || t + P_NW < 9) ++c;
    }
    return 2 * c;
}

template <int N>
__device__ __forceinline__ void vm_wait() {
    static_assert(N >= 0 && N <= 15, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DBG: diagnostic builds: 1 = no tap-validity masking (VK_CONV256_DBG=1; timing only, WRONG results); 4 = stamps around the K loop
// and the whole workgroup (VK_PANEL_STAMPS=<file>, tools/panel_stamps.py)
// TAG 1: second symbol for launches of the two-stream backbone section (see conv_mfma_duo.hip)
// MI: 16-pixel row tiles per wave: 8 (256-pixel tiles) or 9 (288-pixel tiles, halo 48 / 112).  Rows are independent, so the tile
// height does not change a bit of the output; it changes how many rounds a grid takes (res4 at 32 x 800 x 1333: 525 tiles of 256
// = 2.05 rounds on 256 CUs, paid as 3; 467 tiles of 288 = 1.82 rounds, paid as 2 x 9/8), and a step carries 36 instead of 32
// MFMAs per wave over the same barrier, weight reads and DMA.  The launcher picks per launch.
template <int PP, int DBG, int TAG = 0, int MI = 8>
__global__ __launch_bounds__(512, 2) void conv3x3_panel_kernel(PanelK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(MI == 8 || MI == 9, "8 or 9 row tiles per wave");
    static_assert(MI == 8 || !(DBG & 8), "the LDS-staged epilogue is written for 128-row halves");
    constexpr int TILE = 2 * MI * 16;             // pixels per tile
    constexpr int PROWS = PP * 128;               // panel rows
    constexpr int HALO = (PROWS - TILE) / 2;      // 64 or 128 (MI = 9: 48 or 112)
    constexpr int PBYTES = PROWS * 64;
    constexpr int WBASE = 2 * PBYTES;
    // PP = 3 leaves 16 KiB of the CU's LDS unused: 64 bytes of zeros behind the weight ring.  A fragment read of a pixel whose tap
    // falls outside the image is pointed there (two VALU instructions per read: a sign-extended bit-field and a bit-select
    // of the address) instead of zeroing the fragment afterwards (four v_cndmask per read and a compare): round 3, see DESIGN.md 6b.
    constexpr bool ZROW = PP == 3;
    constexpr int ZOFF = WBASE + P_NW * P_WSLOT;

    int bid = blockIdx.x;
    const int nwg = p.m_tiles * p.n_tiles;        // tiles (== gridDim.x unless the launch has a dynamic tail)
    unsigned long st_k0 = 0;
    if constexpr (DBG & 4) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k0)::"memory");
    if constexpr (PP == 3) {
        if (p.tile_ctr && bid >= p.static_tiles) {               // (uniform) one atomic per workgroup, through the spare LDS behind the zero row
            int *mail = reinterpret_cast<int *>(smem + 2 * PP * 128 * 64 + P_NW * P_WSLOT + 128);
            if (threadIdx.x == 0) {
                const unsigned v = atomicAdd(p.tile_ctr, 1u);
                if (v == p.ctr_last) *p.tile_ctr = 0u;
                *mail = p.static_tiles + (int)v;
            }
            __syncthreads();
            bid = __builtin_amdgcn_readfirstlane(*mail);
            if (bid >= nwg) return;
        }
    }
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tile = t_ % p.n_tiles, m_tile = t_ / p.n_tiles;
    const int m0 = m_tile * TILE, n0 = n_tile * 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, j = lane & 15;

    unsigned vm0 = 0, vm1 = 0, vm2 = 0;           // per-lane tap validity, taps 0-2 | 3-5 | 6-8: filled behind the prologue's requests

    // ---- LDS-DMA source state ----
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
    const int pm_first = m0 - HALO + wave * 16 + lrow;        // pixel of panel row (piece q): + q*128
    // 32-bit byte offsets from the (uniform) tensor bases: the launcher checks both tensors are < 4 GiB
    const unsigned wsrc0 = (unsigned)(n0 + wave * 32 + lrow) * (unsigned)p.wrow_bytes + lchunk * 16;   // piece i: + i*16 rows
    const unsigned wstep = 16u * p.wrow_bytes;
    auto req_panel = [&](int cs, int piece) {                 // rows (piece*8 + wave)*16 .. +15 of panel(cs)
        int pf = pm_first;
        asm volatile("" : "+v"(pf));                           // opaque (see x_addr_of)
        // rows outside [0, M) are only ever read by taps the validity mask zeroes: any finite-or-not data
        // will do there, so clamp to a real row instead of branching to a zero page
        const int m = min(max(pf + piece * 128, 0), p.M - 1);
        const unsigned off = (unsigned)m * (unsigned)p.cin_bytes + cs * 64 + lchunk * 16;
        VKP_GLDS16(p.x + off, smem + (cs & 1) * PBYTES + (piece * 8 + wave) * 1024);
    };
    auto req_w = [&](int slot, int cs, int tap, int i) {      // rows (wave*2+i)*16 .. +15 of weight slot `slot`
        const unsigned koff = (unsigned)(tap * p.cstages + cs) * 64;  // K layout = (tap, channel): tap*Cin*2 + cs*64 bytes
        unsigned wb_ = wsrc0;
        asm volatile("" : "+v"(wb_));                          // opaque (see x_addr_of)
        VKP_GLDS16(p.w + (wb_ + i * wstep + koff), smem + WBASE + slot * P_WSLOT + (wave * 2 + i) * 1024);
    };

    // ---- fragment addressing ----
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    [[maybe_unused]] const unsigned zrow = lds0 + ZOFF;
    if constexpr (ZROW) {                         // (complete before the prologue's barrier, ahead of the first fragment read)
        if (tid < 16) *reinterpret_cast<unsigned *>(smem + ZOFF + tid * 4) = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned w_a[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int wrow = wc * 64 + (j >> 2) * 8 + par * 4 + (j & 3);
        w_a[par] = lds0 + WBASE + wrow * 64 + ((g ^ ((-(wrow >> 2)) & 3)) << 4);
    }
    const int jbase = HALO + wr * (MI * 16) + j;              // panel row of fragment pixel mi = 0 at zero shift
    auto x_addr_of = [&](int parity, int tap) -> unsigned {   // byte address of fragment row mi = 0 (panel `parity`, tap)
        const int shift = ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.dil;
        int jb = jbase;
        asm volatile("" : "+v"(jb));                           // opaque: the 18 (panel, tap) addresses must not be hoisted
        const int rb = jb + shift;                             // >= 0: HALO >= dil*(W+1)
        return lds0 + parity * PBYTES + (rb << 6) + ((g ^ ((-(rb >> 2)) & 3)) << 4);
    };

    floatx4 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 wa[4], wb[4], xw[MI == 9 ? 5 : 4];      // (MI = 9: row tile 8 has a fragment register of its own)
    const int CS = p.cstages;

#define VKP_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
    // fragment read of row tile MI_ of tap J_ (TW: the lane's validity word of taps 3*(J_/3)..+2, unshifted).  Written as asm: left to
    // hipcc the select becomes and + compare + s_nop + cndmask.
#define VKP_DSRX(dst, addr, OFF, TW, J_, MI_)                                                        \
    do {                                                                                             \
        if constexpr (ZROW && !(DBG & 1)) {                                                          \
            unsigned m_, ad_;                                                                        \
            const unsigned zc_ = zrow - (OFF);                                                       \
            asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(m_) : "v"(TW), "n"(((J_) % 3) * 9 + (MI_))); \
            asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(ad_) : "v"(m_), "v"(addr), "s"(zc_));    \
            VKP_DSR(dst, ad_, OFF);                                                                  \
        } else                                                                                       \
            VKP_DSR(dst, addr, OFF);                                                                 \
    } while (0)
#define VKP_WAIT3(reg) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(reg))
#define VKP_WAIT4(reg) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(reg))
#define VKP_SB() __builtin_amdgcn_sched_barrier(0)
    // zero a fragment whose pixel is outside the image for this tap (4 v_cndmask), then 4 MFMAs
#define VKP_MMA_ROW(MI, XR, WF, TM)                                                                  \
    do {                                                                                             \
        if constexpr (!(DBG & 1) && !ZROW) {                                                         \
            uintx4 u_ = __builtin_bit_cast(uintx4, XR);                                              \
            const bool v_ = ((TM) >> (MI)) & 1u;                                                     \
            u_[0] = v_ ? u_[0] : 0u;                                                                 \
            u_[1] = v_ ? u_[1] : 0u;                                                                 \
            u_[2] = v_ ? u_[2] : 0u;                                                                 \
            u_[3] = v_ ? u_[3] : 0u;                                                                 \
            XR = __builtin_bit_cast(half8, u_);                                                      \
        }                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[MI][ni] =                               \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[ni], XR, acc[MI][ni], 0, 0, 0);                \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)
    // (the slot addresses are recomputed in the loop from opaque copies: hipcc would otherwise keep all
    //  2 x 6 slot addresses in registers)
#define VKP_READ_W(WF, so)                              \
    do {                                                \
        unsigned a0_ = w_a[0], a1_ = w_a[1];            \
        asm volatile("" : "+v"(a0_), "+v"(a1_));        \
        a0_ += so;                                      \
        a1_ += so;                                      \
        VKP_DSR(WF[0], a0_, 0);                         \
        VKP_DSR(WF[1], a1_, 0);                         \
        VKP_DSR(WF[2], a0_, 2048);                      \
        VKP_DSR(WF[3], a1_, 2048);                      \
    } while (0)

    // One step = tap J of channel stage cs (global step t = 9*cs + J).  LAST: cs is the final stage
    // (no panel prefetch, weight requests stop when t + 6 >= 9*CS, no step after tap 8).
    // ODD = parity of cs (stages run in pairs, so it is known at compile time): panel buffer = ODD, and the
    // weight slot of step t = 9*cs + J is (3*ODD + J) % 6.
    // A step is cut at its barrier into PRE (rows 0-4, then every LDS read of the step is complete) and POST (rows
    // 5-7 together with the first fragment reads of the next step); the loop iterates POST(t-1) + PRE(t), so NO
    // hand-issued ds_read is in flight at a loop boundary, where hipcc may copy fragment registers
    // (tools/check_asm_hazards.py; see conv_mfma256.hip).
    auto tap_mask = [&](auto j_c) -> unsigned {
        constexpr int J = decltype(j_c)::value;
        // opaque copy of the mask word BEFORE the shift: otherwise hipcc hoists either the 72 (tap, row-tile)
        // lane masks (SGPR pairs) or the 9 shifted words out of the loop and spills them
        unsigned tw = J < 3 ? vm0 : (J < 6 ? vm1 : vm2);
        asm volatile("" : "+v"(tw));
        return tw >> ((J % 3) * 9);
    };
    auto tap_word = [&](auto j_c) -> unsigned {       // the unshifted word (VKP_DSRX extracts its bit itself)
        constexpr int J = decltype(j_c)::value;
        unsigned tw = J < 3 ? vm0 : (J < 6 ? vm1 : vm2);
        asm volatile("" : "+v"(tw));
        return tw;
    };
    auto pre = [&](auto last_c, auto odd_c, auto j_c, unsigned xa, unsigned &xa_next, const half8 (&wcur)[4]) {
        constexpr bool LAST = decltype(last_c)::value;
        constexpr int ODD = decltype(odd_c)::value;
        constexpr int J = decltype(j_c)::value;
        const unsigned tm = ZROW ? 0u : tap_mask(j_c);
        [[maybe_unused]] const unsigned tw = ZROW ? tap_word(j_c) : 0u;
        constexpr bool has_next = !(LAST && J == 8);
        // address of the next step's fragments (next tap; next stage's panel after tap 8)
        if constexpr (has_next) xa_next = (J == 8) ? x_addr_of(1 - ODD, 0) : x_addr_of(ODD, J + 1);
        VKP_DSRX(xw[3], xa, 3072, tw, J, 3); VKP_WAIT3(xw[0]); VKP_SB(); VKP_MMA_ROW(0, xw[0], wcur, tm); VKP_SB();
        VKP_DSRX(xw[0], xa, 4096, tw, J, 4); VKP_WAIT3(xw[1]); VKP_SB(); VKP_MMA_ROW(1, xw[1], wcur, tm); VKP_SB();
        VKP_DSRX(xw[1], xa, 5120, tw, J, 5); VKP_WAIT3(xw[2]); VKP_SB(); VKP_MMA_ROW(2, xw[2], wcur, tm); VKP_SB();
        VKP_DSRX(xw[2], xa, 6144, tw, J, 6); VKP_WAIT3(xw[3]); VKP_SB(); VKP_MMA_ROW(3, xw[3], wcur, tm); VKP_SB();
        if constexpr (MI == 9) {
            VKP_DSRX(xw[3], xa, 7168, tw, J, 7); VKP_DSRX(xw[4], xa, 8192, tw, J, 8); VKP_WAIT4(xw[0]); VKP_SB(); VKP_MMA_ROW(4, xw[0], wcur, tm); VKP_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3]), "+v"(xw[MI == 9 ? 4 : 3])::"memory");
        } else {
            VKP_DSRX(xw[3], xa, 7168, tw, J, 7); VKP_WAIT3(xw[0]); VKP_SB(); VKP_MMA_ROW(4, xw[0], wcur, tm); VKP_SB();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
        }
        VKP_SB();
        if constexpr (has_next) {
            // vmcnt: everything older than the pieces issued in the last P_NW-2 steps has landed = weights of step
            // t+1 (and, before tap 0, the next stage's panel); barrier: weight slot (t % 6) is free for step t+6,
            // the other panel buffer is free once the stage ends.
            constexpr int OUT = LAST ? last_w_in_window<J>() : 2 * (P_NW - 2) + panel_in_window<J, PP>();
            vm_wait<OUT>();
            // (DBG 16 / 32, timing only, WRONG results: the barrier on every third tap only / never -- what the per-step barrier costs)
            if constexpr (!(DBG & 48) || ((DBG & 16) && J % 3 == 2)) asm volatile("s_barrier" ::: "memory");
        }
        VKP_SB();
    };
    auto post = [&](auto last_c, auto odd_c, auto j_c, int cs, unsigned xa_next, const half8 (&wcur)[4], half8 (&wnext)[4]) {
        constexpr bool LAST = decltype(last_c)::value;
        constexpr int ODD = decltype(odd_c)::value;
        constexpr int J = decltype(j_c)::value;
        static_assert(!(LAST && J == 8), "the final step has no POST");
        constexpr int SLOT = (3 * ODD + J) % P_NW;            // == t % 6; also the slot of step t + 6
        const unsigned tm = ZROW ? 0u : tap_mask(j_c);
        constexpr int JN = (J + 1) % 9;                   // the next step's tap (tap 0 of the next stage after tap 8)
        [[maybe_unused]] const unsigned twn = ZROW ? tap_word(std::integral_constant<int, JN>{}) : 0u;
        constexpr bool issue_w = !LAST || (J + P_NW < 9);
        constexpr bool issue_p = !LAST && (J < PP);
        constexpr unsigned so = (unsigned)((SLOT + 1) % P_NW) * P_WSLOT;
        VKP_READ_W(wnext, so);
        VKP_DSRX(xw[0], xa_next, 0, twn, JN, 0);
        VKP_SB();
        VKP_MMA_ROW(5, xw[1], wcur, tm);
        VKP_SB();
        if constexpr (issue_p) req_panel(cs + 1, J);
        if constexpr (issue_w) req_w(SLOT, cs + (J + P_NW) / 9, (J + P_NW) % 9, 0);
        VKP_DSRX(xw[1], xa_next, 1024, twn, JN, 1);
        VKP_SB();
        VKP_MMA_ROW(6, xw[2], wcur, tm);
        VKP_SB();
        if constexpr (issue_w) req_w(SLOT, cs + (J + P_NW) / 9, (J + P_NW) % 9, 1);
        VKP_DSRX(xw[2], xa_next, 2048, twn, JN, 2);
        VKP_SB();
        VKP_MMA_ROW(7, xw[3], wcur, tm);
        VKP_SB();
        if constexpr (MI == 9) {
            VKP_MMA_ROW(8, xw[MI == 9 ? 4 : 3], wcur, tm);
            VKP_SB();
        }
    };
    // u = 9*ODD + J numbers the 18 steps of a pair of stages; step u works from fragment set u % 2
    auto wset = [&](auto u_c) -> half8(&)[4] {
        if constexpr (decltype(u_c)::value % 2 == 0)
            return wa;
        else
            return wb;
    };
#define VKP_IC(V) std::integral_constant<int, (V)> {}
    // POST(u-1) + PRE(u) inside the pair that starts at stage cs; LP / LC: is step u-1 / u in the final stage
#define VKP_UNIT(LP, LC, U)                                                                                          \
    post(LP, VKP_IC(((U) - 1) / 9), VKP_IC(((U) - 1) % 9), cs + ((U) - 1) / 9, xan, wset(VKP_IC((U) - 1)), wset(VKP_IC(U)));  \
    xa = xan;                                                                                                        \
    pre(LC, VKP_IC((U) / 9), VKP_IC((U) % 9), xa, xan, wset(VKP_IC(U)));
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;

    // ---- prologue: panel(0), weights of steps 0..5; panel(0) and weights(0) must have landed ----
#pragma unroll
    for (int q2 = 0; q2 < PP; ++q2) req_panel(0, q2);
#pragma unroll
    for (int st = 0; st < P_NW; ++st) {
        req_w(st, 0, st, 0);
        req_w(st, 0, st, 1);
    }
    // ---- per-lane tap validity of its MI fragment pixels (rows wr*MI*16 + mi*16 + j): bit ((tap % 3) * 9 + mi) of word tap / 3 ----
    // Computed HERE, behind the prologue's requests, so that it runs under their HBM latency; and without per-pixel integer
    // division (round 3: the first form, two divisions and 81 tap tests per lane ahead of the requests, was ~1300 VALU
    // instructions per wave = 4.7 us of a 133 us workgroup on Res5 conv2).  The wave's first row is placed in its image by
    // two uniform divisions; a lane's row is at most MI*16 - 1 pixels further: column by a float reciprocal with an exact
    // +-1 correction (operands < 2^24), at most one image wrap when the image has >= MI*16 pixels.  Tap validity is separable:
    // three row tests, three column tests.  Images smaller than a wave's rows keep the division form.
    if (p.HW >= MI * 16 + p.W) {
        const int mbase = m0 + wr * (MI * 16);
        const int rem0 = mbase % p.HW, ho0 = rem0 / p.W, wo0 = rem0 - ho0 * p.W;
        const float rcpw = 1.0f / (float)p.W;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int xcol = wo0 + mi * 16 + j;
            int q = (int)((float)xcol * rcpw), wo = xcol - q * p.W;
            if (wo >= p.W) {
                wo -= p.W;
                ++q;
            }
            if (wo < 0) {
                wo += p.W;
                --q;
            }
            int ho = ho0 + q;
            if (ho >= p.H) ho -= p.H;
            const unsigned c0 = (unsigned)(wo - p.dil) < (unsigned)p.W, c1 = 1u, c2 = (unsigned)(wo + p.dil) < (unsigned)p.W;
            const unsigned cw = mbase + mi * 16 + j < p.M ? (c0 << mi) | (c1 << (9 + mi)) | (c2 << (18 + mi)) : 0u;
            vm0 |= (unsigned)(ho - p.dil) < (unsigned)p.H ? cw : 0u;
            vm1 |= cw;
            vm2 |= (unsigned)(ho + p.dil) < (unsigned)p.H ? cw : 0u;
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wr * (MI * 16) + mi * 16 + j;
            if (m < p.M) {
                const int n_img = m / p.HW;
                const int rem = m - n_img * p.HW;
                const int ho = rem / p.W, wo = rem - ho * p.W;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int hi = ho + (tap / 3 - 1) * p.dil, wi = wo + (tap % 3 - 1) * p.dil;
                    const unsigned ok = ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? 1u : 0u;
                    if (tap < 3)
                        vm0 |= ok << (tap * 9 + mi);
                    else if (tap < 6)
                        vm1 |= ok << ((tap - 3) * 9 + mi);
                    else
                        vm2 |= ok << ((tap - 6) * 9 + mi);
                }
            }
        }
    }

    vm_wait<2 * (P_NW - 1)>();
    asm volatile("s_barrier" ::: "memory");
    unsigned long st_c0 = 0, st_r0 = 0;
    if constexpr (DBG & 4) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0)::"memory");
    unsigned xa = x_addr_of(0, 0), xan = 0;
    VKP_READ_W(wa, 0u);
    {
        [[maybe_unused]] const unsigned tw0 = tap_word(VKP_IC(0));
        VKP_DSRX(xw[0], xa, 0, tw0, 0, 0);
        VKP_DSRX(xw[1], xa, 1024, tw0, 0, 1);
        VKP_DSRX(xw[2], xa, 2048, tw0, 0, 2);
    }
    pre(F_{}, VKP_IC(0), VKP_IC(0), xa, xan, wa);
    // CS is even: stages go in pairs (18 steps, so the roles of the two fragment sets repeat per pair)
    int cs = 0;
    for (; cs + 2 < CS; cs += 2) {
        VKP_UNIT(F_{}, F_{}, 1) VKP_UNIT(F_{}, F_{}, 2) VKP_UNIT(F_{}, F_{}, 3) VKP_UNIT(F_{}, F_{}, 4) VKP_UNIT(F_{}, F_{}, 5)
        VKP_UNIT(F_{}, F_{}, 6) VKP_UNIT(F_{}, F_{}, 7) VKP_UNIT(F_{}, F_{}, 8) VKP_UNIT(F_{}, F_{}, 9) VKP_UNIT(F_{}, F_{}, 10)
        VKP_UNIT(F_{}, F_{}, 11) VKP_UNIT(F_{}, F_{}, 12) VKP_UNIT(F_{}, F_{}, 13) VKP_UNIT(F_{}, F_{}, 14) VKP_UNIT(F_{}, F_{}, 15)
        VKP_UNIT(F_{}, F_{}, 16) VKP_UNIT(F_{}, F_{}, 17)
        // last step of this pair, first step of the next one
        post(F_{}, VKP_IC(1), VKP_IC(8), cs + 1, xan, wb, wa);
        xa = xan;
        pre(F_{}, VKP_IC(0), VKP_IC(0), xa, xan, wa);
    }
    // final pair: its second stage is the LAST one
    VKP_UNIT(F_{}, F_{}, 1) VKP_UNIT(F_{}, F_{}, 2) VKP_UNIT(F_{}, F_{}, 3) VKP_UNIT(F_{}, F_{}, 4) VKP_UNIT(F_{}, F_{}, 5)
    VKP_UNIT(F_{}, F_{}, 6) VKP_UNIT(F_{}, F_{}, 7) VKP_UNIT(F_{}, F_{}, 8) VKP_UNIT(F_{}, T_{}, 9) VKP_UNIT(T_{}, T_{}, 10)
    VKP_UNIT(T_{}, T_{}, 11) VKP_UNIT(T_{}, T_{}, 12) VKP_UNIT(T_{}, T_{}, 13) VKP_UNIT(T_{}, T_{}, 14) VKP_UNIT(T_{}, T_{}, 15)
    VKP_UNIT(T_{}, T_{}, 16) VKP_UNIT(T_{}, T_{}, 17)
    {
        const unsigned tm = ZROW ? 0u : tap_mask(VKP_IC(8));
        VKP_MMA_ROW(5, xw[1], wb, tm);
        VKP_MMA_ROW(6, xw[2], wb, tm);
        VKP_MMA_ROW(7, xw[3], wb, tm);
        if constexpr (MI == 9) VKP_MMA_ROW(8, xw[MI == 9 ? 4 : 3], wb, tm);
    }
    unsigned long st_c1 = 0, st_r1 = 0;
    if constexpr (DBG & 4) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c1), "=s"(st_r1)::"memory");
#undef VKP_UNIT
#undef VKP_IC
#undef VKP_DSR
#undef VKP_DSRX
#undef VKP_WAIT3
#undef VKP_WAIT4
#undef VKP_MMA_ROW
#undef VKP_READ_W
#undef VKP_SB

    // ---- epilogue: + bias (+ residual)(ReLU) -> f16 ----
    // DIRECT (the default since round 2): straight from the accumulator layout -- a lane owns 8 consecutive channels of a pixel
    // (16-byte stores, 64 B per pixel row and instruction), specialised on residual / ReLU / ragged tile like conv_ws.hip; same
    // arithmetic, same bits.  DBG & 8: the first form, through LDS in two 128-row halves so that HBM sees whole 512-B rows
    // (two barriers more, 64 ds_write_b128 + 32 ds_read_b128 per thread and tile).
    if constexpr (!(DBG & 8)) {
        auto epilogue = [&](auto res_c, auto relu_c, auto full_c) {
            constexpr bool RES = decltype(res_c)::value, RELU = decltype(relu_c)::value, FULL = decltype(full_c)::value;
            // row tile outer, channel half inner: the two 64-byte halves of a pixel's 128-byte line (this wave's 64 channels) leave in
            // consecutive instructions (round 3; they used to be nine stores apart)
            const int ch0 = n0 + wc * 64 + g * 8;
            floatx4 bq[2][2];
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                bq[qn][0] = *reinterpret_cast<const floatx4 *>(p.bias + ch0 + qn * 32);
                bq[qn][1] = *reinterpret_cast<const floatx4 *>(p.bias + ch0 + qn * 32 + 4);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const long m = m0 + wr * (MI * 16) + mi * 16 + j;
                half8 rr[2];
                if constexpr (RES) {
                    const long mr = min(m, (long)p.M - 1);
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn) rr[qn] = *reinterpret_cast<const half8 *>(p.res + (mr * p.ldy + ch0 + qn * 32) * 2);
                }
#pragma unroll
                for (int qn = 0; qn < 2; ++qn) {
                    floatx4 x0 = acc[mi][2 * qn] + bq[qn][0], x1 = acc[mi][2 * qn + 1] + bq[qn][1];
                    if constexpr (RES) {
                        x0 += __builtin_convertvector(__builtin_shufflevector(rr[qn], rr[qn], 0, 1, 2, 3), floatx4);
                        x1 += __builtin_convertvector(__builtin_shufflevector(rr[qn], rr[qn], 4, 5, 6, 7), floatx4);
                    }
                    half4 h0 = __builtin_convertvector(x0, half4), h1 = __builtin_convertvector(x1, half4);
                    half8 o = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    if constexpr (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                    if (FULL || m < p.M) *reinterpret_cast<half8 *>(p.y + (m * p.ldy + ch0 + qn * 32) * 2) = o;
                }
            }
        };
        const bool full = m0 + TILE <= p.M;
        auto by_full = [&](auto r_, auto l_) {
            if (full)
                epilogue(r_, l_, std::true_type{});
            else
                epilogue(r_, l_, std::false_type{});
        };
        auto by_relu = [&](auto r_) {
            if (p.relu)
                by_full(r_, std::true_type{});
            else
                by_full(r_, std::false_type{});
        };
        if (p.res)
            by_relu(std::true_type{});
        else
            by_relu(std::false_type{});
    } else {
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
#define VKP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    half8 r0[8];
    load_res(0, r0);
    stage_half(0);
    VKP_LDS_BARRIER();
    write_half(0, r0);
    load_res(1, r0);
    VKP_LDS_BARRIER();
    stage_half(1);
    VKP_LDS_BARRIER();
    write_half(1, r0);
#undef VKP_LDS_BARRIER
    }
    if constexpr (DBG & 4) {
        unsigned long st_k1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k1)::"memory");
        if (threadIdx.x == 0) {
            unsigned long *o = p.stamps + (long)blockIdx.x * 6;
            o[0] = st_c1 - st_c0;       // K loop, core cycles
            o[1] = st_r1 - st_r0;       // K loop, 10 ns ticks
            o[2] = st_k1 - st_k0;       // whole workgroup, 10 ns ticks
            o[3] = st_r0 - st_k0;       // start .. K loop start
            o[4] = st_k0;               // absolute start / end (10 ns ticks): which XCD finishes when
            o[5] = st_k1;
        }
    }
}

static int panel_pp(const ConvArgs &a, int mi = 8) {
    const int reach = a.dil * (a.W + 1);
    if (mi == 9) return reach <= 48 ? 3 : (reach <= 112 ? 4 : 0);
    return reach <= 64 ? 3 : (reach <= 128 ? 4 : 0);
}

// 8 or 9 row tiles per wave (256- or 288-pixel tiles): whichever takes fewer MFMA rows over the rounds of the grid on this device.
// VK_PANEL_MI=8 / 9 forces one where it is legal (A/B switch and bit-identity tests; re-read per call).
static int panel_mi(const ConvArgs &a, long M) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 8;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (panel_pp(a, 9) == 0) return 8;
    if (const char *v = getenv("VK_PANEL_MI")) {
        if (v[0] == '8') return 8;
        if (v[0] == '9') return 9;
    }
    // few rounds: count MFMA rows over the rounds (res4 of 32 images: 3 x 8 against 2 x 9).  Many rounds: the workgroups drift apart
    // and the rounds blur; 36 MFMAs per barrier instead of 32 then win by ~1 % (measured on the Res5 conv2 grids, 14 rounds)
    const long nt = a.Cout / 256;
    const long n8 = ((M + 255) / 256 * nt + n_cu - 1) / n_cu, n9 = ((M + 287) / 288 * nt + n_cu - 1) / n_cu;
    if (n8 > 4) return 9;
    return n9 * 9 < n8 * 8 ? 9 : 8;
}

bool conv3x3_panel_eligible(const ConvArgs &a) {
    const char *v = getenv("VK_CONV3X3_PANEL");          // "0" disables (A/B switch, re-read per call)
    if (v && v[0] == '0') return false;
    if (a.stem || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return false;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != a.dil) return false;
    if (a.Cout % 256 != 0 || a.ldy != a.Cout || a.Cin % 64 != 0 || a.Cin < 128) return false;
    if (panel_pp(a) == 0) return false;
    if ((long)a.N * a.H * a.W * a.Cin * 2 >= (1L << 32)) return false;   // 32-bit DMA offsets
    // no lower bound on M: the K walk (channel-stage major, tap minor) sums in a different order than the
    // im2col kernels, and a layer's bits must not depend on how many RoIs share a launch (head chunking)
    return true;
}

int launch_conv3x3_panel(const ConvArgs &a, hipStream_t stream) {
    PanelK k;
    k.x = (const char *)a.x;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    k.H = a.H;
    k.W = a.W;
    k.HW = a.H * a.W;
    const long M = (long)a.N * a.H * a.W;
    VK_REQUIRE(M > 0 && M < (1L << 31) - 512, VK_EINVAL, "conv3x3_panel: M=%ld out of range", M);
    VK_REQUIRE(M * a.Cin * 2 < (1L << 32) && (long)a.Cout * 9 * a.Cin * 2 < (1L << 32), VK_EINVAL,
               "conv3x3_panel: tensor beyond the 32-bit DMA offsets (M=%ld Cin=%d Cout=%d)", M, a.Cin, a.Cout);
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.ldy = a.ldy;
    k.dil = a.dil;
    k.cstages = a.Cin / 32;
    k.wrow_bytes = 9 * a.Cin * 2;
    k.relu = a.relu;
    const int mi = panel_mi(a, M), pp = panel_pp(a, mi);
    k.m_tiles = ceil_div(k.M, mi * 32);
    k.n_tiles = a.Cout / 256;
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    const int total_tiles = k.m_tiles * k.n_tiles;
    k.tile_ctr = nullptr;
    k.static_tiles = total_tiles;
    int n_wgs = total_tiles;
    const char *dyn_env = getenv("VK_PANEL_DYNAMIC");                        // "0": every tile static (A/B switch and bit-identity test; re-read per call)
    const bool dyn_off = dyn_env && dyn_env[0] == '0';
    if (!dyn_off && pp == 3 && total_tiles >= 16 * 256) {          // many rounds on every CU: the last sixteenth is handed out dynamically
        VK_TRY(acquire_tile_counter(&k.tile_ctr));
        k.static_tiles = total_tiles * 15 / 16 / 8 * 8;             // (a multiple of 8: whole rounds of the XCD map; even: column-tile pairs stay together)
        n_wgs = total_tiles + (total_tiles / 32 + 63) / 64 * 64;    // spare workgroups: an XCD 3 % faster than the mean can take 3 % more tiles
        k.ctr_last = (unsigned)(n_wgs - k.static_tiles - 1);
    }
    const dim3 grid(n_wgs), block(512);
#ifdef VK_ABLATION
    const int dbg = getenv("VK_CONV256_DBG") ? atoi(getenv("VK_CONV256_DBG")) : 0;
#endif
    const int smem = 2 * pp * 128 * 64 + P_NW * P_WSLOT + (pp == 3 ? 256 : 0);      // PP = 3: + the zero row
    k.stamps = nullptr;
#define VKP_LAUNCH(PP_, DBG_, TAG_, MI_)                                                                                                   \
    do {                                                                                                                                   \
        static bool attr_ = false;                                                                                                         \
        if (!attr_) {                                                                                                                      \
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_panel_kernel<PP_, DBG_, TAG_, MI_>),                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PP_ * 128 * 64 + P_NW * P_WSLOT + (PP_ == 3 ? 256 : 0)));            \
            attr_ = true;                                                                                                                  \
        }                                                                                                                                  \
        hipLaunchKernelGGL((conv3x3_panel_kernel<PP_, DBG_, TAG_, MI_>), grid, block, smem, stream, k);                                     \
    } while (0)
#ifdef VK_ABLATION      // stamp / timing-only (dbg 1: WRONG results) / LDS-epilogue builds: tools/ builds only (make ABLATION=1)
    if (const char *sf = getenv("VK_PANEL_STAMPS"); sf && pp == 3) {   // diagnostic: one stamped launch (halo-64 / 48 build), 4 words per workgroup appended to the file
        const size_t nb = (size_t)grid.x * 6 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        VK_CHECK_HIP(hipMemset(k.stamps, 0, nb));                 // (spare workgroups of a dynamic tail leave without a stamp)
        if (mi == 9)
            VKP_LAUNCH(3, 4, 0, 9);
        else
            VKP_LAUNCH(3, 4, 0, 8);
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> h((size_t)grid.x * 6);
        VK_CHECK_HIP(hipMemcpy(h.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# wg k_loop_cycles k_loop_ticks workgroup_ticks ticks_before_k_loop start_tick end_tick (tick = 10 ns)\n");
            for (unsigned w = 0; w < grid.x; ++w)
                fprintf(f, "%u %lu %lu %lu %lu %lu %lu\n", w, h[(size_t)w * 6], h[(size_t)w * 6 + 1], h[(size_t)w * 6 + 2], h[(size_t)w * 6 + 3], h[(size_t)w * 6 + 4],
                        h[(size_t)w * 6 + 5]);
            fclose(f);
        }
    } else if (mi == 8 && pp == 3 && dbg == 8)
        VKP_LAUNCH(3, 8, 0, 8);
    else if (mi == 8 && pp == 3 && dbg == 1)
        VKP_LAUNCH(3, 1, 0, 8);
    else if (mi == 9 && pp == 3 && dbg == 1)
        VKP_LAUNCH(3, 1, 0, 9);
    else if (mi == 9 && pp == 3 && dbg == 16)
        VKP_LAUNCH(3, 16, 0, 9);
    else if (mi == 9 && pp == 3 && dbg == 32)
        VKP_LAUNCH(3, 32, 0, 9);
    else
#endif
    if (mi == 8 && pp == 3 && a.concurrent)
        VKP_LAUNCH(3, 0, 1, 8);
    else if (mi == 8 && pp == 3)
        VKP_LAUNCH(3, 0, 0, 8);
    else if (mi == 8 && a.concurrent)
        VKP_LAUNCH(4, 0, 1, 8);
    else if (mi == 8)
        VKP_LAUNCH(4, 0, 0, 8);
    else if (pp == 3 && a.concurrent)
        VKP_LAUNCH(3, 0, 1, 9);
    else if (pp == 3)
        VKP_LAUNCH(3, 0, 0, 9);
    else if (a.concurrent)
        VKP_LAUNCH(4, 0, 1, 9);
    else
        VKP_LAUNCH(4, 0, 0, 9);
#undef VKP_LAUNCH
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        tm->recs.push_back({a.concurrent ? 6 : 4, 2.0 * (double)k.M * a.Cout * 9 * a.Cin, e0, e1, k.M, a.Cout, a.Cin, 9, 1,
                            2.0 * ((double)k.M * a.Cin + (double)k.M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * 9 * a.Cin)});
    }
    return VK_OK;
}

}  // namespace vk
