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
#include <cstdlib>

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

bool bneck_fused_eligible(int cin, int cmid, int cout, int stride, int groups, bool proj, long N, int H, int W, vk_dtype dt) {
    const char *v = getenv("VK_BNECK_FUSED");            // "0" disables (A/B switch and comparison tests; re-read per call)
    if (v && v[0] == '0') return false;
    if (dt != VK_F16 || cmid != 64 || cout != 256 || stride != 1 || groups != 1) return false;
    if (!((cin == 256 && !proj) || (cin == 64 && proj))) return false;
    if (H < 1 || W < 1 || N * H * W * 512 >= (1L << 31)) return false;       // 32-bit byte offsets into x and y
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
