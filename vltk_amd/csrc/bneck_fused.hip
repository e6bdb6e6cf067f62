// A whole res2 BottleneckBlock as ONE kernel (reference vltk/modeling/frcnn.py:963-979):
//     out = relu(conv3(relu(conv2(relu(conv1(x))))) + shortcut(x))        64 bottleneck channels, 256 out, stride 1
// SURVEY.md §7 "HBM vs MFMA balance": res2 is HBM-bound layer by layer (conv1 reads 512 B per pixel to write 128, conv3
// reads 128 + 512 to write 512, conv2 128 -> 128: 2048 B per pixel and block against 1024 if x is read once and `out` is
// written once).  Here the two 64-channel intermediates never leave the CU:
//   * unit of work = an 8 x 32 tile of output pixels; its 10 x 34 halo of x arrives by LDS-DMA in chunks of 32 pixels through
//     a ring of five slots that runs across tiles (the next tile's first chunks land while this one is multiplied);
//   * phase A: conv1 + BN + ReLU on the 340 halo pixels (zero outside the image: that is conv2's padding) -> t1 in LDS (f16,
//     the same rounding point as the layer-by-layer path);  phase B: the 3x3 conv from t1 (nine taps = nine shifted reads, as
//     conv3x3_blk.hip) -> t2 in LDS;  phase C: conv3 + BN + residual + ReLU -> HBM.  Block 0 (projection shortcut on the 64
//     stem channels, PROJ): conv3 and the shortcut are one GEMM over K = [t2 | x], as in the layer-by-layer path;
//   * all three weight matrices (136 KB) live in REGISTERS for the lifetime of a persistent workgroup: four waves, one per
//     SIMD, 2 (halves of the output channels) x 2 (halves of the pixel blocks); weights cost no memory traffic;
//   * HBM traffic per pixel: 512 B x 340 / 256 read (the halo overlap is served by L2: neighbouring tiles run on the same XCD)
//     + 512 B written; no t1 / t2 / second read of x.
// Same MFMA (v_mfma_f32_16x16x32_f16), same K order and the same epilogue arithmetic as conv_ws.hip / conv3x3_blk.hip.
// Residual reads and output stores are hand-issued (asm) so that their waits can be counted: hipcc waits vmcnt(0) for any load
// it sees while LDS-DMA is in flight.  Lanes whose pixel is outside the image load from a clamped address and store to a
// scratch page, so every wave issues the same number of vector-memory instructions and the static counts hold.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct BneckK {
    const char *x;       // [N, H, W, CIN] f16
    const char *w1;      // packed rows [>= 64][CIN]
    const char *w2;      // packed rows [>= 64][9 * 64]
    const char *w3;      // packed rows [256][64] (identity) or [256][64 | 64] = [conv3 | shortcut] (PROJ)
    const float *b1, *b2, *b3;
    char *y;             // [N, H, W, 256] f16
    char *trash;         // scratch page for the stores of lanes outside the image
    int N, H, W;
    int tiles_x, tiles_y, ntiles;
    unsigned x_last;     // byte offset of the last pixel of x (loads of pixels outside the image are clamped to [0, x_last])
    unsigned long *stamps;   // STAMP builds (make ABLATION=1, VK_BNECK_STAMPS=<file>): 8 words per wave
};

constexpr int BN_TH = 8, BN_TW = 32, BN_PH = 10, BN_PW = 34, BN_NP = BN_PH * BN_PW;   // 340 halo pixels
// ring chunks are 16 KB for both block kinds: 32 halo pixels of 256 channels (11 chunks per tile, ring of 5) or 128 pixels of 64
// channels (3 chunks per tile, ring of 3 = a whole tile ahead)
constexpr int bn_chpx(int cin) { return cin == 256 ? 32 : 128; }
constexpr int bn_ring(int cin) { return cin == 256 ? 5 : 3; }
constexpr int BN_T1_BYTES = BN_NP * 128, BN_T2_BYTES = BN_TH * BN_TW * 128;
constexpr int BN_TRASH_BYTES = 1 << 16;

template <int CIN>
constexpr int bn_smem() { return bn_ring(CIN) * bn_chpx(CIN) * CIN * 2 + BN_T1_BYTES + BN_T2_BYTES + 384 * 4; }   // + the biases

#define VKN_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void bn_vm_wait() {
    static_assert(N >= 0, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");     // fewer than the true count: waits longer, never shorter
}
template <int OFF>
__device__ __forceinline__ void bn_gload(half8 &dst, const char *addr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void bn_gstore(char *addr, half8 val) {
    // s_nop 1: a 128-bit store reads its data registers over the two states after issue; hipcc does not pad an asm statement, and
    // its next instruction may rewrite them (seen: dword 1 of some stores replaced by the next unit's sums)
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" ::"v"(addr), "v"(val), "n"(OFF) : "memory");
}

// CIN: channels of x (256: identity block; 64: block 0).  PROJ: conv3 and a projection shortcut as one GEMM, no residual.
// STAMP: diagnostic build (tools only): core cycles of the three phases summed over a wave's tiles (s_memtime; the stamps
// go to a buffer of their own and no output depends on them)
template <int CIN, bool PROJ, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void bneck64_kernel(BneckK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert((CIN == 256 && !PROJ) || (CIN == 64 && PROJ), "res2 shapes");
    constexpr int BN_CHPX = bn_chpx(CIN), BN_RING = bn_ring(CIN);
    constexpr int BN_NCH = (BN_NP + BN_CHPX - 1) / BN_CHPX;
    constexpr int BPW = BN_CHPX / 32;             // 16-pixel blocks per wave and chunk
    static_assert(BN_NCH >= BN_RING && 2 * BN_NCH > BN_NCH + BN_RING - 1, "the ring reaches at most into the next tile");
    constexpr int PXB = CIN * 2;                  // bytes per pixel of x
    constexpr int CHB = BN_CHPX * PXB;            // ring slot
    constexpr int SPP = PXB / 16;                 // 16-byte slots per pixel (32 / 8)
    constexpr int KM = SPP == 32 ? 15 : 7;        // swizzle key mask: slot' = slot ^ (pixel & KM)
    constexpr int PPI = 64 / SPP;                 // pixels per DMA instruction (2 / 8)
    constexpr int NQ = BN_CHPX / PPI / 4;         // DMA instructions per wave and chunk (4 / 1)
    constexpr int KS1 = CIN / 32;                 // K steps of conv1
    constexpr int KS3 = PROJ ? 4 : 2;             // K steps of conv3 (+ shortcut)
    constexpr int NL = PROJ ? 2 : 4;              // hand-issued loads per block of phase C (x fragments / residual units)
    char *const T1 = smem + BN_RING * CHB;
    char *const T2 = T1 + BN_T1_BYTES;
    float *const B3 = reinterpret_cast<float *>(T2 + BN_T2_BYTES);      // conv3's 256 biases (32 registers per lane otherwise)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cp = wave & 1;                      // which half of the output channels of each phase
    const int par = wave >> 1;                    // which half of the pixel blocks
    const int g = lane >> 4, j = lane & 15;

    // ---- weights and biases: registers, for the lifetime of the workgroup ----
    // row j of MFMA tile ni is output channel (j>>2)*8 + ni*4 + (j&3) of a 32-channel unit, so that a lane ends up with 8
    // consecutive channels (one 16-byte store / LDS write), as in conv_mfma.hip
    const int rowc = (j >> 2) * 8 + (j & 3);
    half8 w1f[2][KS1], w2f[2][9][2], w3f[4][2][KS3];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int co = cp * 32 + rowc + ni * 4;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) w1f[ni][ks] = *reinterpret_cast<const half8 *>(p.w1 + (long)co * PXB + (ks * 32 + g * 8) * 2);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                w2f[ni][tap][ks] = *reinterpret_cast<const half8 *>(p.w2 + (long)co * (9 * 64 * 2) + (tap * 64 + ks * 32 + g * 8) * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c3 = cp * 128 + u * 32 + rowc + ni * 4;
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) w3f[u][ni][ks] = *reinterpret_cast<const half8 *>(p.w3 + (long)c3 * (KS3 * 64) + (ks * 32 + g * 8) * 2);
        }
    }
    B3[tid] = p.b3[tid];                          // biases in LDS: [conv3's 256 | conv1's 64 | conv2's 64] (48 registers per lane otherwise)
    if (tid < 64) {
        B3[256 + tid] = p.b1[tid];
        B3[320 + tid] = p.b2[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on vmcnt counts DMA pieces and hand-issued loads / stores only
    __syncthreads();

    // ---- this workgroup's tiles: the tile list is cut into 8 contiguous ranges (one per XCD: blocks b and b + 8 share one),
    // the workgroups of an XCD walk their range together, so neighbouring tiles (shared halo) meet in one L2 ----
    const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int rlen = (p.ntiles + 7) >> 3;
    const int t_end = min((xcd + 1) * rlen, p.ntiles);
    const int t_first = xcd * rlen + wgx;
    const int tpi = p.tiles_x * p.tiles_y;

    // byte offset (from p.x) of the halo's first pixel (y0 - 1, x0 - 1) of tile t; may be negative
    auto tile_org = [&](int t, int &n, int &y0, int &x0) {
        n = t / tpi;
        const int tr = t - n * tpi;
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        y0 = ty * BN_TH;
        x0 = tx * BN_TW;
    };
    // Everything derived from the lane index is recomputed per tile from an OPAQUE copy (lq): hipcc would otherwise hoist the
    // ~80 loop-invariant per-lane offsets (44 DMA pieces, 11 t1 rows, ...) out of the tile loop and spill the weights.
    int lq = lane;
    // DMA piece q of this wave for chunk c of the tile whose halo starts at byte offset `org`: PPI pixels, lane -> (pixel, slot)
    auto request = [&](int org, int c, int slot) {
        const int lpx = lq / SPP;                 // pixel of the instruction this lane fetches a slot of
        const int lsl = lq % SPP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pic = (wave * NQ + q) * PPI + lpx;          // pixel within the chunk
            const int pp = c * BN_CHPX + pic;                       // halo pixel (>= 340 in the last chunk: any address will do)
            const int pr = pp / BN_PW, pc = pp - pr * BN_PW;
            int off = org + (pr * p.W + pc) * PXB;
            off = min(max(off, 0), (int)p.x_last);                // outside the image: clamped (phase A writes zeros there)
            off += (lsl ^ (pic & KM)) << 4;
            VKN_GLDS16(p.x + (unsigned)off, smem + slot * CHB + (wave * NQ + q) * 1024);
        }
    };

    // ---- prologue: the first tile's first RING chunks ----
    int t = t_first;
    int n = 0, y0 = 0, x0 = 0;
    if (t < t_end) {
        tile_org(t, n, y0, x0);
        const int org = ((n * p.H + y0 - 1) * p.W + x0 - 1) * PXB;
#pragma unroll
        for (int c = 0; c < BN_RING; ++c) request(org, c, c);
    }
    int slot0 = 0;                                // ring slot of this tile's chunk 0
    bool first = true;
    unsigned long st_a = 0, st_b = 0, st_c = 0, st_n = 0, st_t0 = 0, st_r0 = 0;
    auto now = [&]() -> unsigned long {
        unsigned long v = 0;
        if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };
    if constexpr (STAMP) {
        st_t0 = now();
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_r0)::"memory");
    }
    for (; t < t_end; t += nwx) {
        const int org = ((n * p.H + y0 - 1) * p.W + x0 - 1) * PXB;
        const int tn = t + nwx;
        int nn = 0, ny0 = 0, nx0 = 0;
        if (tn < t_end) tile_org(tn, nn, ny0, nx0);
        const int norg = ((nn * p.H + ny0 - 1) * p.W + nx0 - 1) * PXB;
        const bool has_next = tn < t_end;
        asm volatile("" : "+v"(lq));                  // opaque per tile (see above)
        const int g = lq >> 4, j = lq & 15;
        // ---- per-lane LDS offsets ----
        const int a0 = (par * 16 + j) * PXB + ((g ^ (j & KM)) << 4);          // phase A fragment: ^ (ks << 6)
        int colb[3];                                                          // phase B fragment of tap column dx: ^ (ks << 6)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) colb[dx] = (j + dx) * 128 + ((g ^ ((j + dx) & 7)) << 4);
        const int c0 = j * 128 + ((g ^ (j & 7)) << 4);                        // phase C fragment / the lane's 16 B of a T2 row (slot g)
        const int wsl = ((cp * 4 + g) ^ (j & 7)) << 4;                        // T2 write: slot cp*4 + g of pixel j (16-aligned blocks)

        const unsigned long s0 = now();
        // =========================== phase A: t1 = relu(conv1(x) + b1) on the halo ===========================
        // chunk c sits in slot (slot0 + c) % RING; at iteration c: wait for this wave's pieces of chunk c, barrier (publishes the
        // other waves' pieces; every wave is done with chunk c - 1), request chunk c + RING - 1 (of this tile or of the next) into
        // the slot chunk c - 1 has left, multiply chunk c.  Requests in issue order: [chunks 0 .. RING-1 before the tile], then one
        // chunk per iteration c >= 1, so at the wait of iteration c the pieces younger than chunk c's are those of RING - 2 chunks
        // (iteration 0 of the first tile: the prologue's RING - 1).  Chunks 0 .. RING-1 of a LATER tile need no wait: they were
        // requested before the previous tile's phase C, whose hand-issued loads -- younger, and returned in order -- were all
        // waited for there; a wait here would only stall on that phase's stores.
#pragma unroll
        for (int c = 0; c < BN_NCH; ++c) {
            if (c >= BN_RING) {
                bn_vm_wait<(BN_RING - 2) * NQ>();
            } else if (first) {
                if (c == 0)
                    bn_vm_wait<(BN_RING - 1) * NQ>();
                else
                    bn_vm_wait<(BN_RING - 2) * NQ>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            {   // chunk c + RING - 1 into the slot chunk c - 1 has left
                const int cn = c + BN_RING - 1;
                const int sl = (slot0 + cn) % BN_RING;
                if (c >= 1) {
                    if (cn < BN_NCH)
                        request(org, cn, sl);
                    else if (has_next)
                        request(norg, cn - BN_NCH, sl);
                    else
                        request(org, BN_NCH - 1, sl);            // keeps the count of vector-memory instructions the same
                }
            }
            const char *ch = smem + ((slot0 + c) % BN_RING) * CHB;
#pragma unroll
            for (int bw = 0; bw < BPW; ++bw) {       // this wave's blocks of the chunk: par, par + 2, ...
                floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const half8 xf = *reinterpret_cast<const half8 *>(ch + bw * 32 * PXB + (a0 ^ (ks << 6)));
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[0][ks], xf, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[1][ks], xf, acc1, 0, 0, 0);
                }
                // lane: halo pixel pp, channels cp*32 + g*8 .. +8
                const int pp = c * BN_CHPX + (par + 2 * bw) * 16 + j;
                const int pr = pp / BN_PW, pc = pp - pr * BN_PW;
                const bool inside = (unsigned)(y0 - 1 + pr) < (unsigned)p.H && (unsigned)(x0 - 1 + pc) < (unsigned)p.W;
                half8 o;
                const floatx4 ba = *reinterpret_cast<const floatx4 *>(B3 + 256 + cp * 32 + g * 8), bb = *reinterpret_cast<const floatx4 *>(B3 + 260 + cp * 32 + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v0 = acc0[e] + ba[e], v1 = acc1[e] + bb[e];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    o[e] = inside ? (_Float16)v0 : (_Float16)0.f;      // conv2's zero padding
                    o[4 + e] = inside ? (_Float16)v1 : (_Float16)0.f;
                }
                if (pp < BN_NP) *reinterpret_cast<half8 *>(T1 + pp * 128 + (((cp * 4 + g) ^ (pc & 7)) << 4)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // t1 is complete; the last chunk's slot is free
        {
            const int cn = BN_NCH + BN_RING - 1;    // the next tile's chunk RING - 1
            const int sl = (slot0 + cn) % BN_RING;
            if (has_next)
                request(norg, cn - BN_NCH, sl);
            else
                request(org, BN_NCH - 1, sl);
        }

        // Phase C's hand-issued loads run TWO blocks ahead, and the first two blocks' are issued here, ahead of phase B: an L2 /
        // Infinity-Cache round trip is longer than a block of phase C, and with one wave per SIMD nothing else covers it (one
        // block ahead, issued inside phase C: 14.9k cycles per tile for 2k cycles of MFMAs and ~5k of epilogue arithmetic).
        half8 ld[3][NL];
        char *ya[3];
        auto addrs = [&](int i, const char *&xa_, char *&ya_) {
            const int pb = par + 2 * i;
            const int r = pb >> 1, cb = pb & 1;
            const int oy = y0 + r, ox = x0 + cb * 16 + j;
            const bool ok = oy < p.H && ox < p.W;
            const int pix = (n * p.H + oy) * p.W + ox;
            int xo = pix * PXB;
            xo = min(max(xo, 0), (int)p.x_last);
            xa_ = p.x + (unsigned)xo + g * 16 + (PROJ ? 0 : cp * 256);
            ya_ = ok ? p.y + (long)pix * 512 + cp * 256 + g * 16 : p.trash + lq * 16 + cp * 256 + wave * 4096;
        };
        auto issue_loads = [&](int i, half8 (&d)[NL], char *&ya_) {
            const char *xa_;
            addrs(i, xa_, ya_);
            if constexpr (PROJ) {
                bn_gload<0>(d[0], xa_);
                bn_gload<64>(d[1], xa_);
            } else {
                bn_gload<0>(d[0], xa_);
                bn_gload<64>(d[1], xa_);
                bn_gload<128>(d[2], xa_);
                bn_gload<192>(d[3], xa_);
            }
        };
        issue_loads(0, ld[0], ya[0]);
        issue_loads(1, ld[1], ya[1]);
        const unsigned long s1 = now();
        // =========================== phase B: t2 = relu(conv2(t1) + b2), 3x3, nine shifted reads of t1 ===========================
        // this wave's blocks: pb = par + 2 i (i = 0 .. 7), block pb = tile row pb >> 1, columns (pb & 1) * 16 .. + 15; fragment
        // reads run one tap row ahead of the MFMAs in two register sets (conv3x3_blk.hip)
        {
            half8 xa[6], xb[6];
            floatx4 acc0, acc1;
            auto rd = [&](half8 (&xr)[6], int i, int dy) {
                if (i >= 8) return;
                const int pb = par + 2 * i;
                const int r = pb >> 1, cb = pb & 1;
                const char *rowp = T1 + ((r + dy) * BN_PW + cb * 16) * 128;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) xr[dx * 2 + ks] = *reinterpret_cast<const half8 *>(rowp + (colb[dx] ^ (ks << 6)));
            };
            auto mm = [&](const half8 (&xr)[6], int dy) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[0][dy * 3 + dx][ks], xr[dx * 2 + ks], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[1][dy * 3 + dx][ks], xr[dx * 2 + ks], acc1, 0, 0, 0);
                    }
            };
            auto out = [&](int i) {
                const int pb = par + 2 * i;
                half8 o;
                const floatx4 ba = *reinterpret_cast<const floatx4 *>(B3 + 320 + cp * 32 + g * 8), bb = *reinterpret_cast<const floatx4 *>(B3 + 324 + cp * 32 + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v0 = acc0[e] + ba[e], v1 = acc1[e] + bb[e];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    o[e] = (_Float16)v0;
                    o[4 + e] = (_Float16)v1;
                }
                *reinterpret_cast<half8 *>(T2 + pb * 16 * 128 + j * 128 + wsl) = o;
            };
#define VKN_BLOCK(X, Y, i)                                  \
    {                                                       \
        acc0 = floatx4{0.f, 0.f, 0.f, 0.f};                 \
        acc1 = floatx4{0.f, 0.f, 0.f, 0.f};                 \
        rd(Y, i, 1);                                        \
        __builtin_amdgcn_sched_barrier(0);                  \
        mm(X, 0);                                           \
        __builtin_amdgcn_sched_barrier(0);                  \
        rd(X, i, 2);                                        \
        __builtin_amdgcn_sched_barrier(0);                  \
        mm(Y, 1);                                           \
        __builtin_amdgcn_sched_barrier(0);                  \
        rd(Y, (i) + 1, 0);                                  \
        __builtin_amdgcn_sched_barrier(0);                  \
        mm(X, 2);                                           \
        __builtin_amdgcn_sched_barrier(0);                  \
        out(i);                                             \
    }
            rd(xa, 0, 0);
            VKN_BLOCK(xa, xb, 0)
            VKN_BLOCK(xb, xa, 1)
            VKN_BLOCK(xa, xb, 2)
            VKN_BLOCK(xb, xa, 3)
            VKN_BLOCK(xa, xb, 4)
            VKN_BLOCK(xb, xa, 5)
            VKN_BLOCK(xa, xb, 6)
            VKN_BLOCK(xb, xa, 7)
#undef VKN_BLOCK
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // t2 is complete (both channel halves of every pixel)

        const unsigned long s2 = now();
        // =========================== phase C: y = relu(conv3(t2) [+ shortcut(x)] + b3 [+ x]) -> HBM ===========================
        // blocks as in phase B; output channels cp*128 + u*32 .. (u = 0 .. 3).  Per block NL hand-issued loads (identity: the
        // residual's four 16-byte pieces per lane; PROJ: the two x fragments of the shortcut) two blocks ahead, and 4 stores.
        {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                half8(&cur)[NL] = ld[i % 3];
                if (i + 2 < 8) issue_loads(i + 2, ld[(i + 2) % 3], ya[(i + 2) % 3]);
                const int pb = par + 2 * i;
                const half8 t0 = *reinterpret_cast<const half8 *>(T2 + pb * 16 * 128 + c0);
                const half8 t1 = *reinterpret_cast<const half8 *>(T2 + pb * 16 * 128 + (c0 ^ 64));
                // issue order of a wave's phase-C instructions: L0 L1 | L2 S0 | L3 S1 | ... | L7 S5 | S6 | S7 (Li: the NL loads of
                // block i, Si: its 4 stores).  Behind block i's loads when they are needed: nL later load blocks (2; block 6: 1;
                // block 7: 0) and nS earlier store blocks (0, 1, then 2)
                const int nL = i <= 5 ? 2 : 7 - i, nS = i < 2 ? i : 2;
                if constexpr (PROJ) {
                    if (NL * nL + 4 * nS == 4)
                        asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
                    else if (NL * nL + 4 * nS == 8)
                        asm volatile("s_waitcnt vmcnt(8)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
                    else if (NL * nL + 4 * nS == 10)
                        asm volatile("s_waitcnt vmcnt(10)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
                    else
                        asm volatile("s_waitcnt vmcnt(12)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][0], t0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][0], t0, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][1], t1, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][1], t1, acc1, 0, 0, 0);
                    if constexpr (PROJ) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][KS3 - 2], cur[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][KS3 - 2], cur[0], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][KS3 - 1], cur[1], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][KS3 - 1], cur[1], acc1, 0, 0, 0);
                    }
                    const float *bl = B3 + cp * 128 + u * 32 + g * 8;
                    floatx4 v0 = acc0 + *reinterpret_cast<const floatx4 *>(bl), v1 = acc1 + *reinterpret_cast<const floatx4 *>(bl + 4);
                    if constexpr (!PROJ) {
                        half8 &rr = cur[u];
                        // piece u: + the 3 - u pieces behind it + this block's u stores so far = 3 more
                        if (3 + 4 * (nL + nS) == 11)
                            asm volatile("s_waitcnt vmcnt(11)" : "+v"(rr)::"memory");
                        else if (3 + 4 * (nL + nS) == 15)
                            asm volatile("s_waitcnt vmcnt(15)" : "+v"(rr)::"memory");
                        else
                            asm volatile("s_waitcnt vmcnt(19)" : "+v"(rr)::"memory");
                        v0 += __builtin_convertvector(__builtin_shufflevector(rr, rr, 0, 1, 2, 3), floatx4);
                        v1 += __builtin_convertvector(__builtin_shufflevector(rr, rr, 4, 5, 6, 7), floatx4);
                    }
                    const half4 h0 = __builtin_convertvector(v0, half4), h1 = __builtin_convertvector(v1, half4);
                    half8 o = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                    char *ya_ = ya[i % 3];
                    if (u == 0) bn_gstore<0>(ya_, o);
                    if (u == 1) bn_gstore<64>(ya_, o);
                    if (u == 2) bn_gstore<128>(ya_, o);
                    if (u == 3) bn_gstore<192>(ya_, o);
                }
            }
        }
        // (T2 is rewritten only in the next tile's phase B, T1 in its phase A: both behind barriers every wave passes after
        //  finishing this phase)
        if constexpr (STAMP) {
            const unsigned long s3 = now();
            st_a += s1 - s0;
            st_b += s2 - s1;
            st_c += s3 - s2;
            st_n += 1;
        }
        slot0 = (slot0 + BN_NCH) % BN_RING;
        n = nn;
        y0 = ny0;
        x0 = nx0;
        first = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the padding requests of the last tile land before the LDS is released
    if constexpr (STAMP) {
        const unsigned long t1_ = now();
        unsigned long r1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_)::"memory");
        if (lane == 0) {
            unsigned long *o = p.stamps + ((long)blockIdx.x * 4 + wave) * 8;
            o[0] = st_a;
            o[1] = st_b;
            o[2] = st_c;
            o[3] = st_n;
            o[4] = t1_ - st_t0;      // whole kernel, core cycles
            o[5] = r1_ - st_r0;      // whole kernel, 10 ns ticks
        }
    }
}


// =====================================================================================================================
// ROW-STREAMING form (the default): the tile form above re-reads x for the residual and reads a 10 x 34 halo per 8 x 32 outputs:
// 3.6 GB through the fabric per identity block at bench size against 2.2 GB of x-in + y-out, and its time is exactly that
// traffic at ~6 TB/s (stamps: phase A waits for LDS-DMA, phase C for the residual).  Here a unit of work is a COLUMN STRIP:
// <= 30 output columns (+ 1 halo column each side = one 32-pixel row of x = two MFMA blocks) x RU rows, walked row by row:
//   step s:  x halo row s arrives (LDS-DMA, P rows ahead, ring of R = P + 4 rows)
//            A: conv1 on x row s            -> t1 row s        (ring of 4 rows in LDS)
//            B: 3x3 on t1 rows s-3 .. s-1   -> t2 row s-3      (2 rows in LDS)
//            C: conv3 on t2 row s-4 + residual = x row s-3, STILL IN THE RING -> y row s-4 -> HBM
// The three parts of a step only use what earlier steps published: ONE barrier per step, and hipcc is free to interleave
// them.  x is read once (32 / TW columns per output column, two extra rows per unit), nothing else is read: the only
// hand-issued vector-memory instructions are the stores, and every step issues the same number of them (to the scratch page
// while the pipeline fills and drains), so the one counted wait per step is static.
struct BneckRowsK {
    const char *x, *w1, *w2, *w3;
    const float *b1, *b2, *b3;
    char *y, *trash;
    int N, H, W;
    int strips, tw;          // column strips per image and their width (<= 30)
    int vsplit, ru;          // row units per strip and their height
    int nunits;
    unsigned long *stamps;
};

#ifndef VK_BR_P
#define VK_BR_P 3
#endif
constexpr int BR_P = VK_BR_P;                    // x rows in flight ahead of the row being multiplied
constexpr int BR_R = BR_P + 4;                   // x-row ring: row s is last read (residual) at step s + 3
constexpr int BR_T1_ROW = 32 * 128, BR_T1_BYTES = 4 * BR_T1_ROW + 256, BR_T2_BYTES = 2 * 32 * 128;
template <int CIN>
constexpr int br_smem() { return BR_R * 32 * CIN * 2 + BR_T1_BYTES + BR_T2_BYTES + 384 * 4; }

// EIGHT waves (two per SIMD): wave = (cq, par) = (a quarter of the channels, one of the row's two 16-pixel blocks).  With four
// waves (one per SIMD, the tile form's split) a step took ~4000 cycles for 1100 cycles of MFMAs: a wave alone issues one vector
// instruction per 4 cycles, half of the weights sat in AGPRs behind v_accvgpr_read, and nothing covered LDS round trips.  With
// eight, a wave's weights are 136 registers (A and B: ONE 16-channel MFMA tile, C: 64 output channels), the SIMD interleaves
// two instruction streams, and the fragments a wave reads from LDS feed one MFMA instead of two (LDS ~1100 cycles per step).
// DBG (tools build only; WRONG results): timing-only ablations: 1 no A, 2 no B, 4 no C arithmetic, 8 no stores, 16 no DMA
template <int CIN, bool PROJ, bool STAMP = false, int DBG = 0>
__global__ __launch_bounds__(512, 2) void bneck64_rows_kernel(BneckRowsK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert((CIN == 256 && !PROJ) || (CIN == 64 && PROJ), "res2 shapes");
    constexpr int PXB = CIN * 2;                  // bytes per pixel of x
    constexpr int XROW = 32 * PXB;                // one x row of the strip (16 KB / 4 KB)
    constexpr int SPP = PXB / 16;                 // 16-byte slots per pixel (32 / 8)
    constexpr int KM = SPP == 32 ? 15 : 7;        // swizzle key mask: slot' = slot ^ (pixel & KM)
    constexpr int PPI = 64 / SPP;                 // pixels per DMA instruction (2 / 8)
    constexpr int NI = 32 / PPI;                  // DMA instructions per row (16 / 4): wave w issues NI / 8 of them (2; PROJ: waves 0-3 one)
    constexpr int NQ = NI >= 8 ? NI / 8 : 1;
    constexpr int KS1 = CIN / 32;
    constexpr int KS3 = PROJ ? 4 : 2;
    char *const T1 = smem + BR_R * XROW;
    char *const T2 = T1 + BR_T1_BYTES;
    float *const BI = reinterpret_cast<float *>(T2 + BR_T2_BYTES);     // [conv3's 256 | conv1's 64 | conv2's 64] biases

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = wave & 3, par = wave >> 2;
    const bool dma_wave = NI >= 8 || wave < NI;   // PROJ: a row is four DMA instructions
    // ---- weights: registers, for the lifetime of the workgroup ----
    // A, B: MFMA row j = channel cq*16 + j (a lane ends up with 4 consecutive channels: 8 bytes).  C: row j of tile ni of unit u
    // = channel cq*64 + u*32 + (j>>2)*8 + ni*4 + (j&3) (8 consecutive channels per lane: one 16-byte store)
    half8 w1f[KS1], w2f[9][2], w3f[2][2][KS3];
    {
        const int g = lane >> 4, j = lane & 15;
        const int co = cq * 16 + j;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) w1f[ks] = *reinterpret_cast<const half8 *>(p.w1 + (long)co * PXB + (ks * 32 + g * 8) * 2);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) w2f[tap][ks] = *reinterpret_cast<const half8 *>(p.w2 + (long)co * (9 * 64 * 2) + (tap * 64 + ks * 32 + g * 8) * 2);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int c3 = cq * 64 + u * 32 + (j >> 2) * 8 + ni * 4 + (j & 3);
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) w3f[u][ni][ks] = *reinterpret_cast<const half8 *>(p.w3 + (long)c3 * (KS3 * 64) + (ks * 32 + g * 8) * 2);
            }
    }
    if (tid < 256) BI[tid] = p.b3[tid];
    if (tid < 64) {
        BI[256 + tid] = p.b1[tid];
        BI[320 + tid] = p.b2[tid];
    }
    // the row behind the T1 ring that taps of the two padding columns (outputs 30, 31: never stored) read
    if (tid < 16) reinterpret_cast<floatx4 *>(T1 + 4 * BR_T1_ROW)[tid] = floatx4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on vmcnt counts DMA pieces and hand-issued stores only
    __syncthreads();

    unsigned long st_w = 0, st_n = 0, st_t0 = 0, st_r0 = 0;
    auto now = [&]() -> unsigned long {
        unsigned long v = 0;
        if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };
    if constexpr (STAMP) {
        st_t0 = now();
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_r0)::"memory");
    }

    // ---- this workgroup's units: 8 contiguous ranges of the unit list (one per XCD), walked together by its workgroups ----
    const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3, nwx = gridDim.x >> 3;
    const int rlen = (p.nunits + 7) >> 3;
    const int u_end = min((xcd + 1) * rlen, p.nunits);
    int lq = lane;
    for (int unit = xcd * rlen + wgx; unit < u_end; unit += nwx) {
        asm volatile("" : "+v"(lq));                  // per-lane offsets are recomputed per unit (see the tile form)
        const int g = lq >> 4, j = lq & 15;
        // unit -> (image, row unit, strip): strips of one row unit are neighbours in the list (they share halo columns)
        const int strip = unit % p.strips, vu = (unit / p.strips) % p.vsplit, n = unit / (p.strips * p.vsplit);
        const int x0 = strip * p.tw, y0 = vu * p.ru;
        const int tw = min(p.tw, p.W - x0), ru = min(p.ru, p.H - y0);       // this unit's outputs: ru rows x tw columns
        // ---- per-lane offsets ----
        const int pcA = par * 16 + j;                                       // this lane's halo column in A (x / t1 pixel of the row)
        const int aoff = pcA * PXB + ((g ^ (pcA & KM)) << 4);                // A fragment: ^ (ks << 6)
        // t1 / t2 writes: 4 channels cq*16 + g*4 .. = bytes cq*32 + g*8 of the pixel: 16-byte slot cq*2 + (g>>1), half g&1
        const int t1w = pcA * 128 + (((cq * 2 + (g >> 1)) ^ (pcA & 7)) << 4) + (g & 1) * 8;
        const bool colA = (unsigned)(x0 - 1 + pcA) < (unsigned)p.W;         // halo column inside the image
        const int c = par * 16 + j;                                         // this lane's output column in B / C
        int colb[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) colb[dx] = (c + dx) * 128 + ((g ^ ((c + dx) & 7)) << 4);
        const int t2w = c * 128 + (((cq * 2 + (g >> 1)) ^ (c & 7)) << 4) + (g & 1) * 8;
        const int t2r = c * 128 + ((g ^ (c & 7)) << 4);                      // ^ 64 for the second K step
        // x centre pixel of output column c.  Identity: the residual's channels cq*64 + u*32 + g*8 = slot cq*8 + u*4 + g (^ u << 6);
        // PROJ: the shortcut's K step ks = slot ks*4 + g (^ ks << 6)
        const int xr = PROJ ? (c + 1) * PXB + ((g ^ ((c + 1) & 7)) << 4)
                            : (c + 1) * PXB + (cq >> 1) * 256 + ((((cq & 1) * 8 + g) ^ ((c + 1) & 15)) << 4);
        const bool colC = c < tw;                                           // (x0 + c < W follows from tw)
        const long ycol = (long)(x0 + c) * 512 + cq * 128 + g * 16;
        char *const ytrash = p.trash + lq * 16 + wave * 2048;
        // DMA: piece q of this wave covers pixels (wave*NQ + q)*PPI + lane / SPP of the row; the column is clamped into the image
        int dcol[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pic = ((wave * NQ + q) * PPI + lq / SPP) & 31;
            const int col = min(max(x0 - 1 + pic, 0), p.W - 1);
            dcol[q] = col * PXB + (((lq % SPP) ^ (pic & KM)) << 4);
        }
        const long img = (long)n * p.H * p.W;
        auto request = [&](int h) {                  // x halo row h of this unit (image row y0 - 1 + h, clamped) -> ring slot h % R
            if (!dma_wave || (DBG & 16)) return;
            const int yy = min(max(y0 - 1 + h, 0), p.H - 1);
            const unsigned rowb = (unsigned)((img + (long)yy * p.W) * PXB);
            const int sl = h % BR_R;
#pragma unroll
            for (int q = 0; q < NQ; ++q) VKN_GLDS16(p.x + (rowb + (unsigned)dcol[q]), smem + sl * XROW + (wave * NQ + q) * 1024);
        };
#pragma unroll
        for (int h = 0; h < BR_P; ++h) request(h);
        const int nsteps = ru + 4;                   // halo rows 0 .. ru + 1 (A), output rows behind them: B three steps, C four
#pragma clang loop unroll(disable)
        for (int s = 0; s < nsteps; ++s) {
            const unsigned long sw0 = now();
            // row s's pieces: younger are the 2 stores of step s - P and the pieces + stores of steps s - P + 1 .. s - 1; in the first
            // P steps of a unit: the rest of the prologue's rows and the pieces + stores of the steps so far.  (PROJ: waves 4 - 7
            // request nothing and wait for nothing; the barrier publishes the others' rows.)
            static_assert(BR_P >= 2 && BR_P <= 5, "the first-step counts below cover P = 2 .. 5");
            if (dma_wave && !(DBG & 24)) {
                // first steps: (P - 1 - s) prologue rows + s x (pieces + stores); a negative template argument is never instantiated
                if (s >= BR_P)
                    bn_vm_wait<2 + (BR_P - 1) * (NQ + 2)>();
                else if (s == 0)
                    bn_vm_wait<(BR_P - 1) * NQ>();
                else if (s == 1)
                    bn_vm_wait<(BR_P - 2) * NQ + (NQ + 2)>();
                else if (s == 2) {
                    if constexpr (BR_P > 2) bn_vm_wait<(BR_P > 2 ? BR_P - 3 : 0) * NQ + 2 * (NQ + 2)>();
                } else if (s == 3) {
                    if constexpr (BR_P > 3) bn_vm_wait<(BR_P > 3 ? BR_P - 4 : 0) * NQ + 3 * (NQ + 2)>();
                } else {
                    if constexpr (BR_P > 4) bn_vm_wait<4 * (NQ + 2)>();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (STAMP) st_w += now() - sw0;
            request(s + BR_P);                       // (beyond the unit's last halo row: a clamped row nobody reads; same count)

            // The three parts run UNCONDITIONALLY (while the pipeline fills and drains they work on rows nobody reads and C stores to
            // the scratch page), so a step is straight-line code for hipcc to interleave across the parts.
            const char *xrow = smem + (s % BR_R) * XROW;
            const char *t2row = T2 + ((s + 1) & 1) * (32 * 128);       // written at step s - 1
            const char *xres = smem + ((s + BR_R - 3) % BR_R) * XROW;  // x row s - 3: the residual / shortcut input of output row s - 4
            // ---- A: t1 row s = relu(conv1(x row s) + b1), zero outside the image (conv2's padding) ----
            if constexpr (!(DBG & 1)) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[ks], *reinterpret_cast<const half8 *>(xrow + (aoff ^ (ks << 6))), acc, 0, 0, 0);
                const bool inside = colA && (unsigned)(y0 - 1 + s) < (unsigned)p.H;
                const floatx4 ba = *reinterpret_cast<const floatx4 *>(BI + 256 + cq * 16 + g * 4);
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[e] + ba[e];
                    v = v > 0.f ? v : 0.f;
                    o[e] = inside ? (_Float16)v : (_Float16)0.f;
                }
                *reinterpret_cast<half4 *>(T1 + (s & 3) * BR_T1_ROW + t1w) = o;
            }
            // ---- B: t2 row s - 3 = relu(conv2(t1 rows s-3 .. s-1) + b2) ----
            if constexpr (!(DBG & 2)) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const char *rowp = T1 + ((s - 3 + dy) & 3) * BR_T1_ROW;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[dy * 3 + dx][ks], *reinterpret_cast<const half8 *>(rowp + (colb[dx] ^ (ks << 6))), acc, 0, 0, 0);
                }
                const floatx4 ba = *reinterpret_cast<const floatx4 *>(BI + 320 + cq * 16 + g * 4);
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[e] + ba[e];
                    v = v > 0.f ? v : 0.f;
                    o[e] = (_Float16)v;
                }
                *reinterpret_cast<half4 *>(T2 + (s & 1) * (32 * 128) + t2w) = o;
            }
            // ---- C: y row s - 4 = relu(conv3(t2 row s-4) [+ shortcut(x row s-3)] + b3 [+ x row s-3]) -> HBM ----
            {
                const bool live = s >= 4 && s <= ru + 3;               // otherwise: same instructions, stores to the scratch page
                char *yrow = (live && colC) ? p.y + (img + (long)(y0 + s - 4) * p.W) * 512 + ycol : ytrash;
                const half8 t0 = *reinterpret_cast<const half8 *>(t2row + t2r);
                const half8 t1 = *reinterpret_cast<const half8 *>(t2row + (t2r ^ 64));
                half8 xs[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) xs[u] = *reinterpret_cast<const half8 *>(xres + (xr ^ (u << 6)));
                half8 o[2] = {t0, t1};
#pragma unroll
                for (int u = 0; u < ((DBG & 4) ? 0 : 2); ++u) {
                    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][0], t0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][0], t0, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][1], t1, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][1], t1, acc1, 0, 0, 0);
                    if constexpr (PROJ) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][KS3 - 2], xs[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][KS3 - 2], xs[0], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][0][KS3 - 1], xs[1], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3f[u][1][KS3 - 1], xs[1], acc1, 0, 0, 0);
                    }
                    const float *bl = BI + cq * 64 + u * 32 + g * 8;
                    floatx4 v0 = acc0 + *reinterpret_cast<const floatx4 *>(bl), v1 = acc1 + *reinterpret_cast<const floatx4 *>(bl + 4);
                    if constexpr (!PROJ) {
                        v0 += __builtin_convertvector(__builtin_shufflevector(xs[u], xs[u], 0, 1, 2, 3), floatx4);
                        v1 += __builtin_convertvector(__builtin_shufflevector(xs[u], xs[u], 4, 5, 6, 7), floatx4);
                    }
                    const half4 h0 = __builtin_convertvector(v0, half4), h1 = __builtin_convertvector(v1, half4);
                    o[u] = __builtin_elementwise_max(__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), half8{0, 0, 0, 0, 0, 0, 0, 0});
                }
                if constexpr (!(DBG & 8)) {
                    bn_gstore<0>(yrow, o[0]);
                    bn_gstore<64>(yrow, o[1]);
                } else {
                    asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(yrow));
                }
            }
            if constexpr (STAMP) st_n += 1;
        }
        // the next unit's prologue rewrites ring slots 0 .. P-1 and its first steps T1 / T2: every wave must be out of this unit's
        // last step, and this wave's P trailing requests must have landed (they would overwrite the new rows otherwise)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (STAMP) {
        const unsigned long t1_ = now();
        unsigned long r1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_)::"memory");
        if (lane == 0 && wave < 4) {
            unsigned long *o = p.stamps + ((long)blockIdx.x * 4 + wave) * 8;
            o[0] = st_w;             // cycles in the wait + barrier at the head of the steps
            o[1] = 0;
            o[2] = 0;
            o[3] = st_n;             // steps
            o[4] = t1_ - st_t0;
            o[5] = r1_ - st_r0;
        }
    }
}

bool bneck_fused_eligible(int cin, int cmid, int cout, int stride, int groups, bool proj, long N, int H, int W, vk_dtype dt) {
    const char *v = getenv("VK_BNECK_FUSED");            // "0" disables (A/B switch and comparison tests; re-read per call)
    if (v && v[0] == '0') return false;
    if (dt != VK_F16 || cmid != 64 || cout != 256 || stride != 1 || groups != 1) return false;
    if (!((cin == 256 && !proj) || (cin == 64 && proj))) return false;
    // byte offsets into x: unsigned 32-bit in the row form (64 images of 800 x 1333 are 2.2 GB of res2 map), signed in the tile form
    const char *rv = getenv("VK_BNECK_ROWS");
    const long lim = (rv && rv[0] == '0') ? (1L << 31) : (1L << 32) - (1L << 20);
    if (H < 1 || W < 1 || N * H * W * 512 >= lim) return false;
    return true;
}

int launch_bneck_fused(const void *x, int N, int H, int W, int cin, bool proj, const void *w1, const float *b1, const void *w2,
                       const float *b2, const void *w3, const float *b3, void *y, bool concurrent, hipStream_t stream) {
    VK_REQUIRE(bneck_fused_eligible(cin, 64, 256, 1, 1, proj, N, H, W, VK_F16) || getenv("VK_BNECK_FUSED"), VK_EINVAL,
               "bneck_fused: shape not eligible");
    int dev = 0;
    VK_CHECK_HIP(hipGetDevice(&dev));
    VK_REQUIRE(dev >= 0 && dev < VK_MAX_DEVICES, VK_EINVAL, "bneck_fused: device index %d", dev);
    static char *trash[VK_MAX_DEVICES] = {};
    static int n_cu[VK_MAX_DEVICES] = {};
    static bool attr[VK_MAX_DEVICES][2] = {};
    if (!trash[dev]) {
        VK_CHECK_HIP(hipMalloc((void **)&trash[dev], BN_TRASH_BYTES));
        hipDeviceProp_t prop;
        VK_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu[dev] = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
    }
    BneckK k;
    k.x = (const char *)x;
    k.w1 = (const char *)w1;
    k.w2 = (const char *)w2;
    k.w3 = (const char *)w3;
    k.b1 = b1;
    k.b2 = b2;
    k.b3 = b3;
    k.y = (char *)y;
    k.trash = trash[dev];
    k.N = N;
    k.H = H;
    k.W = W;
    k.tiles_x = ceil_div(W, BN_TW);
    k.tiles_y = ceil_div(H, BN_TH);
    const long nt = (long)N * k.tiles_x * k.tiles_y;
    VK_REQUIRE(nt > 0 && nt < (1L << 31), VK_EINVAL, "bneck_fused: %ld tiles", nt);
    k.ntiles = (int)nt;
    k.x_last = (unsigned)(((long)N * H * W - 1) * cin * 2);
    const char *rv = getenv("VK_BNECK_ROWS");            // "0": the tile form (A/B switch and comparison tests; re-read per call)
    if (!(rv && rv[0] == '0')) {
        BneckRowsK r;
        r.x = (const char *)x;
        r.w1 = (const char *)w1;
        r.w2 = (const char *)w2;
        r.w3 = (const char *)w3;
        r.b1 = b1;
        r.b2 = b2;
        r.b3 = b3;
        r.y = (char *)y;
        r.trash = trash[dev];
        r.N = N;
        r.H = H;
        r.W = W;
        r.tw = ceil_div(W, ceil_div(W, 30));
        r.strips = ceil_div(W, r.tw);
        // rows per unit: the split of the image height that takes the fewest steps over the rounds of the grid (a unit of ru rows
        // takes ru + 4 steps and a prologue; at least 16 rows per unit)
        const int grid_ = n_cu[dev];
        long best = -1;
        r.vsplit = 1;
        for (int vs = 1; vs <= std::max(1, H / 16); ++vs) {
            const int ru_ = ceil_div(H, vs), vs_ = ceil_div(H, ru_);
            const long units_ = (long)N * r.strips * vs_;
            const long cost = ((units_ + grid_ - 1) / grid_) * (ru_ + 8);
            if (best < 0 || cost < best) {
                best = cost;
                r.vsplit = vs_;
            }
        }
        r.ru = ceil_div(H, r.vsplit);
        r.vsplit = ceil_div(H, r.ru);
        const long nu = (long)N * r.strips * r.vsplit;
        VK_REQUIRE(nu > 0 && nu < (1L << 31), VK_EINVAL, "bneck_fused: %ld units", nu);
        r.nunits = (int)nu;
        r.stamps = nullptr;
        static bool rattr[VK_MAX_DEVICES][2] = {};
        KernelTimer *tmr = g_timer;
        hipEvent_t f0 = nullptr, f1 = nullptr;
        if (tmr) {
            f0 = tmr->get();
            f1 = tmr->get();
            VK_CHECK_HIP(hipEventRecord(f0, stream));
        }
#ifdef VK_ABLATION
        if (const char *sf = getenv("VK_BNECK_STAMPS")) {     // diagnostic: one stamped launch, 8 words per wave appended to the file
            const size_t nb = (size_t)grid_ * 4 * 8 * sizeof(unsigned long);
            VK_CHECK_HIP(hipMalloc((void **)&r.stamps, nb));
            VK_CHECK_HIP(hipMemsetAsync(r.stamps, 0, nb, stream));
            const int dbg = getenv("VK_BNECK_DBG") ? atoi(getenv("VK_BNECK_DBG")) : 0;
            if (proj) {
                VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_rows_kernel<64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, br_smem<64>()));
                hipLaunchKernelGGL((bneck64_rows_kernel<64, true, true>), dim3(grid_), dim3(512), br_smem<64>(), stream, r);
            } else {
                switch (dbg) {
#define VKN_DBG_CASE(D_)                                                                                                                   \
    case D_:                                                                                                                               \
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_rows_kernel<256, false, true, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, br_smem<256>())); \
        hipLaunchKernelGGL((bneck64_rows_kernel<256, false, true, D_>), dim3(grid_), dim3(512), br_smem<256>(), stream, r);                 \
        break;
                    VKN_DBG_CASE(1) VKN_DBG_CASE(2) VKN_DBG_CASE(4) VKN_DBG_CASE(8) VKN_DBG_CASE(16) VKN_DBG_CASE(7) VKN_DBG_CASE(15) VKN_DBG_CASE(31) VKN_DBG_CASE(24) VKN_DBG_CASE(3)
#undef VKN_DBG_CASE
                    default:
                        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_rows_kernel<256, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, br_smem<256>()));
                        hipLaunchKernelGGL((bneck64_rows_kernel<256, false, true>), dim3(grid_), dim3(512), br_smem<256>(), stream, r);
                }
            }
            VK_CHECK_HIP(hipStreamSynchronize(stream));
            std::vector<unsigned long> hst((size_t)grid_ * 4 * 8);
            VK_CHECK_HIP(hipMemcpy(hst.data(), r.stamps, nb, hipMemcpyDeviceToHost));
            VK_CHECK_HIP(hipFree(r.stamps));
            if (FILE *f = fopen(sf, "a")) {
                fprintf(f, "# rows form: wg wave wait_cycles 0 0 steps kernel_cycles kernel_ticks(10ns)  proj=%d units=%d ru=%d tw=%d\n", (int)proj, r.nunits, r.ru, r.tw);
                for (int w = 0; w < grid_ * 4; ++w) {
                    fprintf(f, "%d %d", w / 4, w % 4);
                    for (int i = 0; i < 6; ++i) fprintf(f, " %lu", hst[(size_t)w * 8 + i]);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
            return VK_OK;
        }
#endif
        if (proj) {
            if (!rattr[dev][1]) {
                VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_rows_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, br_smem<64>()));
                rattr[dev][1] = true;
            }
            hipLaunchKernelGGL((bneck64_rows_kernel<64, true>), dim3(grid_), dim3(512), br_smem<64>(), stream, r);
        } else {
            if (!rattr[dev][0]) {
                VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_rows_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, br_smem<256>()));
                rattr[dev][0] = true;
            }
            hipLaunchKernelGGL((bneck64_rows_kernel<256, false>), dim3(grid_), dim3(512), br_smem<256>(), stream, r);
        }
        VK_CHECK_HIP(hipGetLastError());
        if (tmr) {
            VK_CHECK_HIP(hipEventRecord(f1, stream));
            const double M = (double)N * H * W;
            const double fl = 2.0 * M * (64.0 * cin + 64.0 * 576 + 256.0 * (proj ? 128 : 64));
            tmr->recs.push_back({concurrent ? 6 : 11, fl, f0, f1, (int)M, 256, cin, 3, 1,
                                 2.0 * (M * cin + M * 256 + 64.0 * cin + 64.0 * 576 + 256.0 * (proj ? 128 : 64))});
        }
        return VK_OK;
    }
    VK_REQUIRE((long)N * H * W * 512 < (1L << 31), VK_EINVAL, "bneck_fused: the tile form keeps signed 32-bit byte offsets");
    const int grid = n_cu[dev];                   // a multiple of 8: every XCD walks its own range of tiles
    k.stamps = nullptr;
#ifdef VK_ABLATION
    if (const char *sf = getenv("VK_BNECK_STAMPS")) {     // diagnostic: one stamped launch, 8 words per wave appended to the file
        const size_t nb = (size_t)grid * 4 * 8 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        VK_CHECK_HIP(hipMemsetAsync(k.stamps, 0, nb, stream));
        if (proj) {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_kernel<64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bn_smem<64>()));
            hipLaunchKernelGGL((bneck64_kernel<64, true, true>), dim3(grid), dim3(256), bn_smem<64>(), stream, k);
        } else {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_kernel<256, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bn_smem<256>()));
            hipLaunchKernelGGL((bneck64_kernel<256, false, true>), dim3(grid), dim3(256), bn_smem<256>(), stream, k);
        }
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> hst((size_t)grid * 4 * 8);
        VK_CHECK_HIP(hipMemcpy(hst.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# wg wave phaseA phaseB phaseC tiles kernel_cycles kernel_ticks(10ns)  proj=%d\n", (int)proj);
            for (int w = 0; w < grid * 4; ++w) {
                fprintf(f, "%d %d", w / 4, w % 4);
                for (int i = 0; i < 6; ++i) fprintf(f, " %lu", hst[(size_t)w * 8 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
        return VK_OK;
    }
#endif
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    if (proj) {
        if (!attr[dev][1]) {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bn_smem<64>()));
            attr[dev][1] = true;
        }
        hipLaunchKernelGGL((bneck64_kernel<64, true>), dim3(grid), dim3(256), bn_smem<64>(), stream, k);
    } else {
        if (!attr[dev][0]) {
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bneck64_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bn_smem<256>()));
            attr[dev][0] = true;
        }
        hipLaunchKernelGGL((bneck64_kernel<256, false>), dim3(grid), dim3(256), bn_smem<256>(), stream, k);
    }
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const double M = (double)N * H * W;
        const double fl = 2.0 * M * (64.0 * cin + 64.0 * 576 + 256.0 * (proj ? 128 : 64));
        tm->recs.push_back({concurrent ? 6 : 11, fl, e0, e1, (int)M, 256, cin, 3, 1,
                            2.0 * (M * cin + M * 256 + 64.0 * cin + 64.0 * 576 + 256.0 * (proj ? 128 : 64))});
    }
    return VK_OK;
}

}  // namespace vk
