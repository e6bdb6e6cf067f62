// 1x1 stride-1 convolution with a LONG K (>= 1024 channels, one or two inputs) as a 256x256-tile GEMM with FOUR waves of
// 128 x 128: conv1 of the Res5 bottlenecks (1024 / 2048 -> 512, reference vltk/modeling/frcnn.py:955-960) and conv3 + projection
// shortcut as one GEMM (K = 512 | 1024 -> 2048, :970-977).
//
// Why another GEMM body: under an MFMA-dense loop the chip holds 1.5 - 1.9 GHz and what it delivers follows the energy per
// MFMA, not the stalls.  The 8-wave ring kernel (conv_mfma256.hip, 128 x 64 per wave) reads 12 LDS fragments
// per 32 MFMAs; with 128 x 128 per wave it is 16 per 64 -- a third fewer LDS bytes per MFMA, the shape the vendor's own GEMM
// uses on these sizes (1134 - 1220 TFLOP/s against the ring kernel's 975 - 1000).  Everything else is the ring kernel's design:
//   * K advances in stages of 32 channels; LDS is a ring of 4 stage slots of (256 pixel rows + 256 channel rows) x 64 B filled by
//     LDS-DMA (global_load_lds_dwordx4, XOR swizzle on the source address), pixels four stages ahead, weights three;
//   * one raw barrier per stage after row 4 of 8: counted vmcnt (stage s+1 landed, two stages in flight), lgkmcnt(0), barrier;
//   * fragment reads issued by hand three pixel rows ahead of the 8 MFMAs that consume them (counted lgkmcnt), the next stage's
//     8 weight fragments under rows 5-7; one DMA piece after each row's MFMAs, where its issue cost sits under the matrix pipe;
//   * one wave per SIMD, 512 registers per lane: 256 accumulator registers, two sets of 8 weight fragments, a 4-deep window of
//     pixel-row fragments.
// The epilogue stores straight from the accumulator layout (a lane owns 8 consecutive channels of a pixel: 16-byte stores,
// 64 B per pixel row and instruction), specialised on residual / ReLU / ragged tile like conv_ws.hip.  Same K order and
// epilogue arithmetic as the other 1x1 kernels: a layer's bits do not depend on which one the dispatcher picks.
#include <cstdio>
#include <type_traits>
#include <vector>

#include <atomic>
#include <mutex>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct Gemm4K {
    const char *x;       // [M, cin] rows
    const char *x2;      // second K segment (stages >= st1): [M, cin2] rows, or nullptr
    const char *w;       // packed weight rows [Cout][cin (+ cin2)]
    const float *bias;
    const char *res;
    char *y;
    int M;
    int cin_bytes, cin2_bytes;
    int xmul, x2mul;     // row pitches beyond the descriptor's 14-bit stride: stride = pitch / mul, index = row * mul (1, 2 or 4)
    int st1;             // 32-channel stages of the first input (== stages when x2 is null)
    int stages;          // all stages (even, >= 8)
    int wrow_bytes;
    int ldy;
    int relu;
    int m_tiles, n_tiles;
    unsigned long *stamps;   // STAMP builds only: 4 words per workgroup
    // Dynamic tail (round 3): tiles [0, static_tiles) are walked bid, bid + grid, ... as before; the rest are handed out by an atomic
    // counter, whichever workgroup asks first.  The XCDs of one chip differ by 4 - 6 % in speed under the power limit (stamps:
    // tools/gemm4_stamps.py), and with a fixed share per workgroup the fast ones sat idle for 3.4 % of the K = 2048 launches.
    unsigned *tile_ctr;      // nullptr: every tile is static
    int static_tiles;
    unsigned ctr_last;       // the launch's last fetch (every workgroup fetches until its first miss): whoever draws it zeroes the word for its next user
};

constexpr int G_ROWB = 64, G_NSLOT = 4;
constexpr int G_XB = 256 * G_ROWB;           // 16 KiB
constexpr int G_SLOT = 2 * G_XB;             // 32 KiB
constexpr int G_SMEM = G_NSLOT * G_SLOT;     // 128 KiB (the ring; the layer's bias, Cout x 4 B, sits behind it)

// STAMP: diagnostic build (VK_GEMM4_STAMPS=<file>): wave 0 stamps s_memtime / s_memrealtime around the first tile's K loop and
// sums the phases of all its tiles into a buffer nothing else reads (tools/gemm4_stamps.py); never used by the product path
// DBG (STAMP builds, timing only, WRONG results; VK_GEMM4_DBG): 1 = no LDS-DMA in the steady state, 2 = no stage barrier there,
// 4 = no pixel-row fragment reads there, 8 = no weight fragment reads there; 128 (plain build, correct results): MFMAs through the
// builtin instead of asm -- the reference the asm form was debugged against (hipcc then shuffles accumulators inside the loop)
struct Gemm4Tile {                      // what depends on the tile: origin, the lane's row indices, the three descriptors
    int m0, n0;
    unsigned xi[4];
    __amdgpu_buffer_rsrc_t rx1, rx2, rw;
};

// TAG 1: the same code under a second symbol for launches of the two-stream backbone section (res4's conv1 half-batches), so that
// profilers list the launches that overlap another stream apart from the ones that run alone (see conv_mfma_duo.hip)
template <bool STAMP, int DBG = 0, int TAG = 0>
__global__ __launch_bounds__(256, 1) void conv_gemm4_kernel(Gemm4K p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // PERSISTENT workgroups, one per CU (the launcher sizes the grid), each walking tiles bid, bid + grid, ... as ONE stream of
    // K stages: the ring does not stop at a tile boundary -- the last four stages of a tile request the first stages of the next
    // one (S % 4 == 0, so the slots line up), and a tile's epilogue sits between two MFMA rows with nothing to wait for.
    // (A workgroup per tile paid, per tile, the first stage's HBM latency, the store tail and a dispatch: 14 % of its time.)
    // XCD-aware (bijective) tile map, column tiles of one row tile next to each other on one XCD (conv_mfma.hip); the grid is
    // a multiple of 8 (or the whole tile count), so a workgroup's tiles all map to its own XCD.
    const int bid = blockIdx.x;
    const int total_tiles = p.m_tiles * p.n_tiles;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, j = lane & 15;

    // RULE OF THE K LOOP: no compiler-generated VALU instruction inside it.  The MFMAs are asm (below), so hipcc does not know
    // that a VGPR it sees as dead right after the last MFMA of a row is still being read by the matrix pipe: an address
    // temporary allocated there (v_cndmask / v_add for a DMA piece) corrupted the operand -- wrong rows, found by the bit-identity
    // tests.  Every per-lane address is therefore a loop-invariant register, what advances per stage is scalar, and the LDS
    // slot is a compile-time constant of the (4 x unrolled) loop body folded into the instructions' immediate offsets.
    // Scalars the loop needs are pinned in SGPRs (hipcc otherwise re-loads kernel arguments inside the loop, and the
    // lgkmcnt(0) it then needs drains the hand-counted fragment reads).
    int st1 = p.st1, S = p.stages;
    asm volatile("" : "+s"(st1), "+s"(S));

    // ---- LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane): this lane feeds rows (wave*4 + i)*16 + (lane>>2), i = 0..3, of
    // both blocks of a slot.  Pixel rows: STRUCTURED buffer, index = tile row (constant per lane), the row pitch is the
    // descriptor's stride -- so the two inputs of a dual-source layer differ only in the (scalar) descriptor; offset = the
    // swizzled 16-byte chunk (constant per lane); the K advance is the scalar offset.  Weight rows: raw buffer, 32-bit offset.
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);   // logical 16-B chunk whose bytes land at phys chunk lane&3
    unsigned wv[4];
    const unsigned xo = lchunk * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = (unsigned)((wave * 4 + i) * 16 + lrow) * (unsigned)p.wrow_bytes + lchunk * 16;
    auto setup_tile = [&](int tt, Gemm4Tile &T) {
        const int nwg = total_tiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = tt & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tt >> 3);
        const int n_tile = t % p.n_tiles, m_tile = t / p.n_tiles;
        T.m0 = m_tile * 256;
        T.n0 = n_tile * 256;
        int stid = threadIdx.x;                             // opaque copy: nothing per-lane has to stay alive across the K loop for this
        asm volatile("" : "+v"(stid));
        const int slrow = (stid & 63) >> 2;
#pragma unroll
        // (the index is shared by both inputs, so a layer with two of them needs xmul == x2mul: the launcher checks)
        for (int i = 0; i < 4; ++i) T.xi[i] = (unsigned)min((wave * 4 + i) * 16 + slrow, p.M - 1 - T.m0) * (unsigned)p.xmul;   // rows past M are computed and dropped
        T.rx1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (long)T.m0 * p.cin_bytes), (short)(p.cin_bytes / p.xmul), 0x7fffffff, 0x00020000);
        T.rx2 = __builtin_amdgcn_make_buffer_rsrc((void *)((p.x2 ? p.x2 : p.x) + (long)T.m0 * p.cin2_bytes),
                                                  (short)(p.x2 ? p.cin2_bytes / p.x2mul : p.cin_bytes / p.xmul), 0x7fffffff, 0x00020000);
        T.rw = __builtin_amdgcn_make_buffer_rsrc((void *)(p.w + (long)T.n0 * p.wrow_bytes), 0, 0x7fffffff, 0x00020000);
    };
    const int dma_x0 = (wave * 4) * 1024;                  // byte offset of this wave's first pixel-row piece in a slot
    const int dma_w0 = G_XB + (wave * 4) * 1024;
    bool steady = false;                                    // DBG builds: inside the FULL loop
    // stage = the K stage (of tile T) the piece belongs to, slot = its ring slot (compile time in the loop body)
    auto req_x = [&](const Gemm4Tile &T, int stage, int slot, int i) {
        if ((DBG & 1) && steady) return;
        const bool second = stage >= st1;                   // uniform
        __builtin_amdgcn_struct_ptr_buffer_load_lds(second ? T.rx2 : T.rx1, (__attribute__((address_space(3))) void *)(smem + slot * G_SLOT + dma_x0 + i * 1024),
                                                    16, T.xi[i], xo, (second ? stage - st1 : stage) * G_ROWB, 0, 0);
    };
    auto req_w = [&](const Gemm4Tile &T, int stage, int slot, int i) {
        if ((DBG & 1) && steady) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(T.rw, (__attribute__((address_space(3))) void *)(smem + slot * G_SLOT + dma_w0 + i * 1024), 16, wv[i],
                                                 stage * G_ROWB, 0, 0);
    };

    // ---- fragment read addresses: two bases per operand (slots 0-1 / slots 2-3); the slot's 32 KiB and the row's offset go
    // into the 16-bit immediate ----
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)smem;
    const int sx = (-(j >> 2)) & 3;
    unsigned xa[2], wa0[2], wa1[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        xa[hb] = lds0 + hb * 2 * G_SLOT + (wr * 128 + j) * G_ROWB + ((g ^ sx) << 4);                       // + mi*1024
        const int wrow0 = wc * 128 + (j >> 2) * 8 + (j & 3), wrow1 = wrow0 + 4;                             // + (ni>>1)*32 rows
        wa0[hb] = lds0 + hb * 2 * G_SLOT + G_XB + wrow0 * G_ROWB + ((g ^ ((-(wrow0 >> 2)) & 3)) << 4);
        wa1[hb] = lds0 + hb * 2 * G_SLOT + G_XB + wrow1 * G_ROWB + ((g ^ ((-(wrow1 >> 2)) & 3)) << 4);
    }
    asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(wa0[0]), "+v"(wa0[1]), "+v"(wa1[0]), "+v"(wa1[1]));   // six registers, not re-derived in the loop

    // the layer's bias goes to LDS once (behind the ring): an epilogue that loaded it from memory would wait, in order, behind
    // the next tile's requests
    float *bias_lds = reinterpret_cast<float *>(smem + G_SMEM);
    for (int c = tid; c < p.ldy; c += 256) bias_lds[c] = p.bias[c];

    floatx4 acc[8][8];
    // Fragment registers: two weight sets (current / next stage) and a 4-deep rotating window of pixel-row fragments.  The
    // reads and their COUNTED waits are issued by hand; LDS returns in order.
    half8 wa[8], wb[8], xw[4];

    // ds_read_b128 of fragment row `OFF` bytes into slot SLOT's block (base register pair picked by SLOT >> 1)
#define VKG_DSR(dst, BASE, SLOT, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(BASE[(SLOT) >> 1]), "n"(((SLOT) & 1) * G_SLOT + (OFF)))
#define VKG_DSRX(dst, SLOT, OFF, FULLV)                             \
    do {                                                            \
        if constexpr ((DBG & 4) && FULLV)                           \
            asm volatile("" : "+v"(dst) : "v"(xa[0]));              \
        else                                                        \
            VKG_DSR(dst, xa, SLOT, OFF);                            \
    } while (0)
#define VKG_SB() __builtin_amdgcn_sched_barrier(0)
    // The MFMAs are written as asm with the accumulator tied to an AGPR operand: left to the builtin, hipcc spread the 256
    // accumulator registers over both halves of the register file and moved them around inside the loop (160 v_accvgpr moves
    // per 128 MFMAs).  An accumulator is touched once per stage (64 MFMAs apart), so there is no dependent back-to-back pair.
#define VKG_MF(MI, XR, WF, NI)                                                                                        \
    do {                                                                                                              \
        if constexpr (DBG & 128)                                                                                      \
            acc[MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[NI], XR, acc[MI][NI], 0, 0, 0);                   \
        else                                                                                                          \
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[MI][NI]) : "v"(WF[NI]), "v"(XR));        \
    } while (0)
    // One wave per SIMD: nothing but this wave's own instruction stream feeds the matrix pipe, and an MFMA occupies it for 16
    // cycles, so everything else is issued INSIDE a row, in the gaps after MFMAs 2, 4, 6 and 8 (a row written as 8 MFMAs
    // followed by its fragment read and DMA piece ran 1510 cycles per stage against 1024 of MFMAs; stamps: tools/gemm4_stamps.py).
#define VKG_ROW(MI, XR, WF, G0, G1, G2, G3)  \
    do {                                     \
        VKG_MF(MI, XR, WF, 0);               \
        VKG_MF(MI, XR, WF, 1);               \
        VKG_SB();                            \
        G0;                                  \
        VKG_SB();                            \
        VKG_MF(MI, XR, WF, 2);               \
        VKG_MF(MI, XR, WF, 3);               \
        VKG_SB();                            \
        G1;                                  \
        VKG_SB();                            \
        VKG_MF(MI, XR, WF, 4);               \
        VKG_MF(MI, XR, WF, 5);               \
        VKG_SB();                            \
        G2;                                  \
        VKG_SB();                            \
        VKG_MF(MI, XR, WF, 6);               \
        VKG_MF(MI, XR, WF, 7);               \
        VKG_SB();                            \
        G3;                                  \
        VKG_SB();                            \
    } while (0)
#define VKG_WAIT2(reg) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(reg))
#define VKG_NONE ((void)0)
#define VKG_IC(V) std::integral_constant<int, (V)> {}

    Gemm4Tile A, B;                          // this tile and the workgroup's next one
    int has_next = 0;                        // workgroup-uniform scalar: this tile is not the workgroup's last

    // The K loop is cut into PRE(s) = rows 0-4 of stage s, ending at the in-stage barrier, and POST(s) = rows 5-7 of stage s
    // together with the first fragment reads of stage s+1; a loop iteration is POST + PRE four times (slots 0..3), so every loop
    // boundary / branch sits right after an `s_waitcnt lgkmcnt(0)`: no hand-issued ds_read is in flight where hipcc may insert
    // register copies (tools/check_asm_hazards.py).
    //   fragment reads: before row r (0..4) exactly x_r, x_r+1, x_r+2 are outstanding -> lgkmcnt(2) = x_r is here; x_r+3 is read
    //     in the row's first gap into the register row r-1 has just been consumed from.  Row 5's gaps read the next stage's 8
    //     weight fragments, rows 6 / 7 the next stage's rows 0-2 (each after the row that consumed its register).
    //   DMA pieces: PRE(s) issues the LAST pixel piece of stage s+3 in row 0 and the four weight pieces of stage s+3 in rows 1-4;
    //     POST(s) the first three pixel pieces of stage s+4 (into the slot PRE(s)'s barrier freed).  Newer than the last piece of
    //     stage s+1 at PRE(s)'s wait: the 8 pieces of stage s+2 and of stage s+3 -- and, just after a tile boundary, the previous
    //     tile's stores: they only make the wait longer (loads and stores do not retire in order with each other, so no count
    //     that had to EXCLUDE them would be safe).
    // SL = s & 3, REM = S - s (stages left including this one; 99 = steady state) and CONT (the workgroup has a next tile:
    // stages beyond this tile's are the next tile's first ones) are compile-time constants: which requests and waits exist is
    // decided per copy of the code, so there is no run-time flag (hipcc turns those into VALU compares).
    auto pre = [&](auto rem_c, auto sl_c, int s, const half8 (&wcur)[8]) {
        constexpr int REM = decltype(rem_c)::value;
        constexpr bool FULL = REM == 99;
        constexpr int SL = decltype(sl_c)::value, S3 = (SL + 3) & 3;
        // stage s+3: of this tile, or of the next one (its stage 3 - REM)
#define VKG_RQX3() do { if constexpr (REM > 3) req_x(A, s + 3, S3, 3); else req_x(B, 3 - REM, S3, 3); } while (0)
#define VKG_RQW(I) do { if constexpr (REM > 3) req_w(A, s + 3, S3, I); else req_w(B, 3 - REM, S3, I); } while (0)
        VKG_WAIT2(xw[0]); VKG_SB();
        VKG_ROW(0, xw[0], wcur, VKG_DSRX(xw[3], SL, 3072, FULL), VKG_RQX3(), VKG_NONE, VKG_NONE);
        VKG_WAIT2(xw[1]); VKG_SB();
        VKG_ROW(1, xw[1], wcur, VKG_DSRX(xw[0], SL, 4096, FULL), VKG_RQW(0), VKG_NONE, VKG_NONE);
        VKG_WAIT2(xw[2]); VKG_SB();
        VKG_ROW(2, xw[2], wcur, VKG_DSRX(xw[1], SL, 5120, FULL), VKG_RQW(1), VKG_NONE, VKG_NONE);
        VKG_WAIT2(xw[3]); VKG_SB();
        VKG_ROW(3, xw[3], wcur, VKG_DSRX(xw[2], SL, 6144, FULL), VKG_RQW(2), VKG_NONE, VKG_NONE);
        VKG_WAIT2(xw[0]); VKG_SB();
        VKG_ROW(4, xw[0], wcur, VKG_DSRX(xw[3], SL, 7168, FULL), VKG_RQW(3), VKG_NONE, VKG_NONE);
#undef VKG_RQX3
#undef VKG_RQW
        // every read of stage s is issued; wait for them in straight-line code
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3])::"memory");
        VKG_SB();
        if constexpr (FULL && (DBG & 2))
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else                                   // (also in a tile's last stages: the stream goes on, two stages in flight behind stage s+1)
            asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        VKG_SB();
    };
    auto post = [&](auto rem_c, auto sl_c, int s, const half8 (&wcur)[8], half8 (&wnext)[8]) {
        constexpr int REM = decltype(rem_c)::value;
        constexpr bool FULL = REM == 99;
        constexpr int SL = decltype(sl_c)::value, SN = (SL + 1) & 3;          // this stage's slot (== the slot of stage s+4), the next stage's
#define VKG_RQX(I) do { if constexpr (REM > 4) req_x(A, s + 4, SL, I); else req_x(B, 4 - REM, SL, I); } while (0)
#define VKG_W2(A_, B_, OFF)                                                     \
    do {                                                                        \
        if constexpr ((DBG & 8) && FULL) {                                      \
            asm volatile("" : "+v"(wnext[A_]), "+v"(wnext[B_]) : "v"(wa0[0]));  \
        } else {                                                                \
            VKG_DSR(wnext[A_], wa0, SN, OFF);                                   \
            VKG_DSR(wnext[B_], wa1, SN, OFF);                                   \
        }                                                                       \
    } while (0)
        VKG_ROW(5, xw[1], wcur, VKG_W2(0, 1, 0), VKG_W2(2, 3, 2048), VKG_W2(4, 5, 4096), VKG_W2(6, 7, 6144));
        VKG_ROW(6, xw[2], wcur, VKG_DSRX(xw[0], SN, 0, FULL), VKG_RQX(0), VKG_DSRX(xw[1], SN, 1024, FULL), VKG_RQX(1));
        VKG_ROW(7, xw[3], wcur, VKG_DSRX(xw[2], SN, 2048, FULL), VKG_RQX(2), VKG_NONE, VKG_NONE);
#undef VKG_W2
#undef VKG_RQX
    };
    // ---- the workgroup's first tile: stages 0..2 completely and the first three pixel pieces of stage 3 (S >= 8 and S % 4 == 0,
    // the launcher checks); every later tile finds exactly this state, left by its predecessor's last four stages ----
    setup_tile(bid, A);
#pragma unroll
    for (int st = 0; st < 3; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) req_x(A, st, st, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) req_w(A, st, st, i);
    }
    req_x(A, 3, 3, 0);
    req_x(A, 3, 3, 1);
    req_x(A, 3, 3, 2);
    asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // stage 0 (the 8 oldest of 27 pieces) has landed; bias is in LDS
    VKG_DSR(wa[0], wa0, 0, 0);
    VKG_DSR(wa[1], wa1, 0, 0);
    VKG_DSR(wa[2], wa0, 0, 2048);
    VKG_DSR(wa[3], wa1, 0, 2048);
    VKG_DSR(wa[4], wa0, 0, 4096);
    VKG_DSR(wa[5], wa1, 0, 4096);
    VKG_DSR(wa[6], wa0, 0, 6144);
    VKG_DSR(wa[7], wa1, 0, 6144);
    VKG_DSR(xw[0], xa, 0, 0);
    VKG_DSR(xw[1], xa, 0, 1024);
    VKG_DSR(xw[2], xa, 0, 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[0]), "+v"(xw[1]), "+v"(xw[2])::"memory");   // (as after a tile boundary: see below)

    bool first_tile = true;
    unsigned long ph_loop = 0, ph_epi = 0, ph_tiles = 0, ts = 0, ts_first = 0, te_last = 0, t_c0 = 0, t_r0 = 0;   // STAMP
    if constexpr (STAMP) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_first)::"memory");
    for (int tt = bid;;) {
        int nt = tt + (int)gridDim.x;
        if (p.tile_ctr && nt >= p.static_tiles) {            // (uniform) the tail: one atomic per workgroup and tile, through LDS
            int *mail = reinterpret_cast<int *>(smem + G_SMEM + p.ldy * 4);
            if (tid == 0) {
                const unsigned v = atomicAdd(p.tile_ctr, 1u);
                if (v == p.ctr_last) *p.tile_ctr = 0u;
                *mail = p.static_tiles + (int)v;
            }
            __syncthreads();
            nt = __builtin_amdgcn_readfirstlane(*mail);
        }
        // (computed by scalar asm: left to hipcc, the flag lives as a lane mask and every branch on it costs a VALU pair -- which it
        // places right behind the MFMAs of the K loop)
        asm volatile("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(has_next) : "s"(nt), "s"(total_tiles) : "scc");
        // B = the workgroup's next tile; on its last tile the stream simply requests this tile's first stages once more (nobody
        // reads them): the K loop and its tail then hold no run-time condition at all
        setup_tile(has_next ? nt : tt, B);
        // zero the accumulators HERE (pinned: hipcc otherwise sinks the zeroing to just before each accumulator's first asm MFMA and,
        // not seeing a matrix instruction there, leaves out the wait states a v_accvgpr write needs before an MFMA reads it as SrcC)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) asm volatile("" : "+a"(acc[mi][ni]));
        asm volatile("s_nop 7" ::: "memory");
        if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_c0), "=s"(t_r0)::"memory");

        pre(VKG_IC(99), VKG_IC(0), 0, wa);
        int s = 0;
        steady = true;
        for (; s + 8 <= S; s += 4) {           // four stages per iteration: the slots are compile-time constants
            post(VKG_IC(99), VKG_IC(0), s, wa, wb);
            pre(VKG_IC(99), VKG_IC(1), s + 1, wb);
            post(VKG_IC(99), VKG_IC(1), s + 1, wb, wa);
            pre(VKG_IC(99), VKG_IC(2), s + 2, wa);
            post(VKG_IC(99), VKG_IC(2), s + 2, wa, wb);
            pre(VKG_IC(99), VKG_IC(3), s + 3, wb);
            post(VKG_IC(99), VKG_IC(3), s + 3, wb, wa);
            pre(VKG_IC(99), VKG_IC(0), s + 4, wa);
        }
        steady = false;
        // s == S - 4, its rows 0-4 done: the last four stages; the requests riding on their rows are the next tile's first stages,
        // and the last POST reads the next tile's first fragments.
        post(VKG_IC(4), VKG_IC(0), s, wa, wb);
        pre(VKG_IC(3), VKG_IC(1), s + 1, wb);
        post(VKG_IC(3), VKG_IC(1), s + 1, wb, wa);
        pre(VKG_IC(2), VKG_IC(2), s + 2, wa);
        post(VKG_IC(2), VKG_IC(2), s + 2, wa, wb);
        pre(VKG_IC(1), VKG_IC(3), s + 3, wb);
        post(VKG_IC(1), VKG_IC(3), s + 3, wb, wa);
        // the epilogue is compiler-scheduled code: no hand-issued read may be in flight across it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[0]), "+v"(xw[1]), "+v"(xw[2])::"memory");
        if constexpr (STAMP) {
            unsigned long t_c1, t_r1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_c1), "=s"(t_r1)::"memory");
            if (tid == 0 && first_tile) {
                unsigned long *o = p.stamps + (long)bid * 8;
                o[0] = t_c1 - t_c0;
                o[1] = t_r1 - t_r0;
            }
            ph_loop += t_r1 - t_r0;
            ts = t_r1;
            ph_tiles += 1;
        }

        // ---- epilogue: (acc + bias) (+ residual) (ReLU) -> f16, straight from the accumulator layout ----
        // hipcc does not see the asm MFMAs as matrix instructions: left alone it starts reading accumulators (v_accvgpr_read for
        // the epilogue, hoisted into the last stage) one instruction after the MFMA that produces their final value -- stale
        // results, found by the bit-identity tests.  So every accumulator is RE-DEFINED by a fence statement that follows the last
        // MFMA's 4 passes + write-back: nothing can read it earlier.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            asm volatile("" : "+a"(acc[mi][0]), "+a"(acc[mi][1]), "+a"(acc[mi][2]), "+a"(acc[mi][3]), "+a"(acc[mi][4]), "+a"(acc[mi][5]),
                         "+a"(acc[mi][6]), "+a"(acc[mi][7]));
        // lane coordinates re-derived from an opaque copy of the thread index: nothing per-lane of the epilogue has to stay alive
        // (or be spilled) across the K loop, whose register file is full
        int etid = threadIdx.x;
        asm volatile("" : "+v"(etid));
        const int eg = (etid & 63) >> 4, ej = etid & 15;
        const int em0 = A.m0, en0 = A.n0;
        auto epilogue = [&](auto res_c, auto relu_c, auto full_c) {
            constexpr bool RES = decltype(res_c)::value, RELU = decltype(relu_c)::value, FULL = decltype(full_c)::value;
            // pixel-row tile outer, 32-channel group inner: a pixel's 256 bytes (this wave's 128 channels, two 128-byte lines) leave
            // in four consecutive stores.  (Group outer put the halves of a line 8 stores apart: WRITE_SIZE 1.08-1.13 x the output.)
            const int ch0 = en0 + wc * 128 + eg * 8;
            floatx4 bq[4][2];
#pragma unroll
            for (int qn = 0; qn < 4; ++qn) {
                bq[qn][0] = *reinterpret_cast<const floatx4 *>(bias_lds + ch0 + qn * 32);
                bq[qn][1] = *reinterpret_cast<const floatx4 *>(bias_lds + ch0 + qn * 32 + 4);
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const long m = em0 + wr * 128 + mi * 16 + ej;
                half8 rr[4];
                if constexpr (RES) {          // (these loads retire in order behind the next tile's requests: a layer with a residual waits for them)
                    const long mr = min(m, (long)p.M - 1);
#pragma unroll
                    for (int qn = 0; qn < 4; ++qn) rr[qn] = *reinterpret_cast<const half8 *>(p.res + (mr * p.ldy + ch0 + qn * 32) * 2);
                }
#pragma unroll
                for (int qn = 0; qn < 4; ++qn) {
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
        const bool full = em0 + 256 <= p.M;
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
        if constexpr (STAMP) {
            unsigned long te;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te)::"memory");
            ph_epi += te - ts;
            te_last = te;
        }
        if (!has_next) break;
        A = B;
        tt = nt;
        first_tile = false;
    }       // tiles of this workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the requests that followed the last tile must have landed before the LDS is released
    if constexpr (STAMP) {
        if (tid == 0) {
            unsigned long *o = p.stamps + (long)bid * 8;
            o[2] = ts_first;
            o[3] = te_last;
            o[4] = 0;
            o[5] = ph_loop;
            o[6] = ph_epi;
            o[7] = ph_tiles;
        }
    }
}

// A zeroed 32-bit device word for ONE launch's dynamic tile tail, from a ring of 256 words per device zeroed when it is allocated.
// No reset on the stream (a memset between two kernels cost an 11 us bubble per launch): the number of fetches of a launch is known
// in advance, and the workgroup that draws the last one writes the zero back.  (A slot comes round again after 256 launches WITH a tail; a handle keeps at most
// four forwards of seven such launches in flight, and the wrap is exercised by every bench run.)
int acquire_tile_counter(unsigned **ctr) {
    static unsigned *ring[VK_MAX_DEVICES];
    static std::atomic<unsigned> next[VK_MAX_DEVICES];
    static std::mutex mu;
    int dev = 0;
    VK_CHECK_HIP(hipGetDevice(&dev));
    VK_REQUIRE(dev >= 0 && dev < VK_MAX_DEVICES, VK_EINVAL, "device %d beyond VK_MAX_DEVICES", dev);
    if (!ring[dev]) {
        std::lock_guard<std::mutex> lock(mu);
        if (!ring[dev]) {
            unsigned *r = nullptr;
            VK_CHECK_HIP(hipMalloc((void **)&r, 256 * sizeof(unsigned)));
            VK_CHECK_HIP(hipMemset(r, 0, 256 * sizeof(unsigned)));      // (synchronous: done before any launch can use a word)
            ring[dev] = r;
        }
    }
    *ctr = ring[dev] + (next[dev].fetch_add(1) & 255u);
    return VK_OK;
}

bool conv_gemm4_eligible(const ConvArgs &a) {
    const char *v = getenv("VK_CONV_GEMM4");             // "0" disables, "2" also takes small grids (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return false;
    const bool any_grid = v && v[0] == '2';
    if (a.stem || a.pool_part || a.groups > 1 || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return false;
    if (a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride != 1) return false;
    if (a.Cout % 256 != 0 || a.ldy != a.Cout || a.Cout > 8192) return false;      // (the bias rides behind the ring in LDS)
    const int cin2 = a.x2 ? a.Cin2 : 0;
    int min_k = 1024;
    if (const char *mk = getenv("VK_GEMM4_MINK")) min_k = atoi(mk) >= 256 ? atoi(mk) : 1024;     // A/B switch (tools/conv_bench.py)
    if (a.Cin % 32 != 0 || cin2 % 32 != 0 || (a.Cin + cin2) % 128 != 0 || a.Cin + cin2 < min_k) return false;   // groups of four 32-channel stages
    // the row pitch is a 14-bit descriptor stride; longer rows (the FPN box head's 12544-wide fc1) run with stride = pitch / 2 or / 4
    // and index = row * 2 or * 4 (one input only: the index is shared)
    if (cin2 ? (a.Cin * 2 >= 16384 || cin2 * 2 >= 16384) : (a.Cin * 2 >= 4 * 16384 || (a.Cin * 2) % 64 != 0)) return false;
    if ((long)a.Cout * (a.Cin + cin2) * 2 >= (1L << 32)) return false;   // 32-bit weight offsets
    const long M = (long)a.N * a.Ho * a.Wo;
    if (M < 8 * 256 || M >= (1L << 31) - 256) return false;
    // measured (end of round 2): ahead of the two-per-CU kernel from one tile per CU on -- res4's conv1, 1024 -> 256 at M = 134 400
    // (525 tiles): 102.5 vs 113.7 us alone, backbone 14.49 -> 14.19 ms in the forward (four interleaved pairs); the first form of
    // this kernel needed four rounds (117 vs 113 us)
    return any_grid || ((M + 255) / 256) * (a.Cout / 256) >= 256;
}

int launch_conv_gemm4(const ConvArgs &a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_gemm4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM + 32768));
#ifdef VK_ABLATION
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_gemm4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM + 32768));
#endif
        attr_set = true;
    }
    const int cin2 = a.x2 ? a.Cin2 : 0;
    Gemm4K k;
    k.x = (const char *)a.x;
    k.x2 = (const char *)a.x2;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    const long M = (long)a.N * a.Ho * a.Wo;
    k.M = (int)M;
    k.cin_bytes = a.Cin * 2;
    k.cin2_bytes = cin2 * 2;
    k.xmul = k.cin_bytes < 16384 ? 1 : (k.cin_bytes < 2 * 16384 ? 2 : 4);
    k.x2mul = 1;
    VK_REQUIRE(k.xmul == 1 || !a.x2, VK_EINVAL, "conv_gemm4: a two-input layer needs rows below 16 KiB");
    k.stages = (a.Cin + cin2) / 32;
    k.st1 = a.x2 ? a.Cin / 32 : k.stages;
    k.wrow_bytes = (a.Cin + cin2) * 2;
    k.ldy = a.ldy;
    k.relu = a.relu;
    k.m_tiles = (int)((M + 255) / 256);
    k.n_tiles = a.Cout / 256;
    VK_REQUIRE(k.stages >= 8 && k.stages % 4 == 0, VK_EINVAL, "conv_gemm4: K = %d", a.Cin + cin2);

    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    k.stamps = nullptr;
    // persistent workgroups, one per CU; a multiple of 8 so that a workgroup's tiles (bid, bid + grid, ...) stay on its XCD
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        VK_CHECK_HIP(hipGetDevice(&dev));
        VK_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
    }
    const int total_tiles = k.m_tiles * k.n_tiles;
    const int grid_wgs = total_tiles < n_cu ? total_tiles : n_cu;
    k.tile_ctr = nullptr;
    k.static_tiles = total_tiles;
    const char *dyn_env = getenv("VK_GEMM4_DYNAMIC");                        // "0": every tile static (A/B switch and bit-identity test; re-read per call)
    const bool dyn_off = dyn_env && dyn_env[0] == '0';
    if (!dyn_off && total_tiles >= 16 * grid_wgs && a.Cout <= 4096) {     // long launches: the last eighth of a workgroup's tiles is dynamic
        VK_TRY(acquire_tile_counter(&k.tile_ctr));
        k.static_tiles = (total_tiles / grid_wgs) * 7 / 8 * grid_wgs;
        k.ctr_last = (unsigned)(total_tiles - k.static_tiles + grid_wgs - 1);     // the tail's tiles + one miss per workgroup
    }
    const size_t smem_bytes = G_SMEM + (size_t)a.Cout * 4 + 16;           // ring, bias, the tail's mailbox word
#ifdef VK_ABLATION      // stamp / timing-only builds (WRONG results for DBG != 0): tools/ builds only (make ABLATION=1)
    if (const char *sf = getenv("VK_GEMM4_STAMPS")) {    // diagnostic: one stamped launch, 4 words per workgroup appended to the file
        const int nwg = grid_wgs;
        const size_t nb = (size_t)nwg * 8 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        const char *d = getenv("VK_GEMM4_DBG");
        switch (d ? atoi(d) : 0) {
#define VKG_DBG_CASE(D_)                                                                                                   \
    case D_:                                                                                                               \
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_gemm4_kernel<true, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM + 32768)); \
        hipLaunchKernelGGL((conv_gemm4_kernel<true, D_>), dim3(nwg), dim3(256), smem_bytes, stream, k);                        \
        break;
            VKG_DBG_CASE(1) VKG_DBG_CASE(2) VKG_DBG_CASE(3) VKG_DBG_CASE(4) VKG_DBG_CASE(8) VKG_DBG_CASE(12) VKG_DBG_CASE(13) VKG_DBG_CASE(15)
#undef VKG_DBG_CASE
            default: hipLaunchKernelGGL(conv_gemm4_kernel<true>, dim3(nwg), dim3(256), smem_bytes, stream, k);
        }
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> h((size_t)nwg * 8);
        VK_CHECK_HIP(hipMemcpy(h.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# wg loop_cycles loop_realtime_ticks stages end_realtime\n");
            for (int w = 0; w < nwg; ++w) {
                fprintf(f, "%d", w);
                for (int i = 0; i < 8; ++i) fprintf(f, " %lu", h[(size_t)w * 8 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    } else if (getenv("VK_GEMM4_DBG") && atoi(getenv("VK_GEMM4_DBG")) == 128) {      // bisect: builtin MFMAs
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_gemm4_kernel<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM + 32768));
        hipLaunchKernelGGL((conv_gemm4_kernel<false, 128>), dim3(grid_wgs), dim3(256), smem_bytes, stream, k);
    } else
#endif
    if (a.concurrent) {
        static bool attr1 = false;
        if (!attr1) {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_gemm4_kernel<false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM + 32768));
            attr1 = true;
        }
        hipLaunchKernelGGL((conv_gemm4_kernel<false, 0, 1>), dim3(grid_wgs), dim3(256), smem_bytes, stream, k);
    } else {
        hipLaunchKernelGGL(conv_gemm4_kernel<false>, dim3(grid_wgs), dim3(256), smem_bytes, stream, k);
    }
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const int K = a.Cin + cin2;
        tm->recs.push_back({a.concurrent ? 6 : 10, 2.0 * (double)M * a.Cout * K, e0, e1, (int)M, a.Cout, K, 1, 1,
                            2.0 * ((double)M * K + (double)M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * K)});
    }
    return VK_OK;
}

}  // namespace vk
