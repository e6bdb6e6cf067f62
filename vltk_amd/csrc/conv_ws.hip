// 1x1 convolution with K <= 512 input channels as a WEIGHT-STATIONARY GEMM: the conv3 layers of the bottlenecks
// (`out = conv3(out) + shortcut`, reference vltk/modeling/frcnn.py:970-979: 512 -> 2048 in the Res5 head, 256 -> 1024 in
// res4, 128 -> 512 in res3, 64 -> 256 in res2), res2's first conv3 with its projection shortcut as one GEMM (64 + 64 channels
// from two tensors), the strided 1x1 convs of res3 / res4 (K = 256 / 512), and the last Res5 conv3 with the spatial mean fused
// (see the template's comments for the three variants).
//
// Why: these layers are bound by what a CU can move through its vector-memory path (~21 B/clk measured), not by
// the matrix cores and not by HBM.  A 128 x 256 tile of the two-per-CU kernel (conv_mfma_duo.hip) moves, per output element
// at K = 512, 4 B of pixels + 8 B of WEIGHTS + 4 B of residual / output; the weights are the same 256 x K block for every
// tile of a column, re-streamed from L2 each time.  Here a workgroup OWNS one 256-channel column block for its whole life:
//   * its weights (256 x K f16 = 256 KiB at K = 512) are loaded once into REGISTERS -- each of the eight waves (two per
//     SIMD, <= 256 registers each) keeps the fragments of its 32 channels for every K step (32 x 4 VGPRs) -- so the K loop
//     streams pixels only: 8 B per output element instead of 16;
//   * pixels run through a 6-slot LDS ring of 64 rows x 128 channels (16 KiB) filled by LDS-DMA five stages ahead,
//     straight across tile boundaries (tiles are 64 rows: 32 accumulator registers beside the 128 of the weights), so the
//     next tile's pixels arrive under this tile's epilogue; one counted vmcnt and one raw barrier per stage; the tile's
//     RESIDUAL rows come by LDS-DMA too (requested when the tile starts, each wave its own channels), so the epilogue
//     has no load to wait for;
//   * workgroups that share an XCD (blockIdx mod 8) and an M lane walk the same pixel tiles with different column blocks,
//     so a pixel tile is fetched from HBM once per XCD L2 and read from L2 by the other column blocks.
// Same K order and epilogue arithmetic as conv_mfma_duo.hip / conv_mfma256.hip ((acc + bias) + residual, ReLU, round to f16):
// a layer's bits do not depend on which of the three the dispatcher picks.
#include "vk_common.h"

#include <cstdio>
#include <type_traits>
#include <vector>

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

struct WsK {
    const char *x;
    const char *x2;          // DUAL builds: the second 64 channels of the K = 128 stage (conv3 + projection shortcut of res2)
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int M;
    int kbytes;              // K * 2 (row pitch of x and of the packed weight rows)
    int ldy;                 // = Cout
    int relu;
    int m_tiles, n_tiles;
    unsigned long *stamps;   // DBG & 32 builds only: 8 cycle sums per wave
    double *pool;            // POOL builds: per-image column sums [M / HW][Cout] (exact in fp64), y is not written
    int HW;                  // rows per image (>= 64: a 64-row tile touches at most two images)
    int H, W, Wo, HoWo, stride;   // STRIDED builds: input geometry of a strided 1x1 conv (x rows are input pixels, M counts output pixels)
};

#define VKW_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void ws_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int WS_BM = 64;                    // rows per tile

// KC = K / 32 MFMA steps (4, 8, 16)
// DBG: timing-only ablation builds (VK_WS_DBG, WRONG results): 1 no epilogue, 2 no MFMA, 4 no pixel DMA, 8 no residual DMA, 64 epilogue without its stores
// NW: waves per workgroup.  4 = one wave per SIMD with 64 channels each (256 weight registers); 8 = two waves per SIMD with
// 32 channels each (128 weight registers, <= 256 registers per lane), so that one wave's barrier / LDS / epilogue latencies
// are covered by the other wave of the SIMD
// Tried on top of the two-waves-per-SIMD form and removed (bit-identical, no faster): FREE-RUNNING waves -- no barrier in the
// loop; a wave announced that ITS pieces of a stage had landed with an LDS atomic add on the stage's arrival counter and
// started a stage after a bounded spin on that counter, so that it could run up to A stages ahead of the slowest wave
// (16 KiB stages, 7 slots, A = 1 / 2: 650 / 622 us vs 647 with the barrier on the 512 -> 2048 layer at M = 200 704; 8 KiB
// stages, 15 slots, A = 4 / 6: 662 / 676 us).  The two waves of a SIMD would have to run a whole TILE apart for one's
// epilogue to sit under the other's MFMAs, and the ring cannot hold that much (LDS: 64 rows x 512 channels = 64 KiB per tile).
// POOL (K = 512, two waves per SIMD): the fused spatial mean of the last Res5 conv3 (`res5(x).mean(dim=[2,3])`, frcnn.py:1401).  The
// f16 values a separate mean kernel would read back are summed per image in fp64 -- EXACTLY: an f16 is a multiple of 2^-24 below
// 2^16, so a sum of up to 2^13 of them fits 53 bits whatever the order -- per lane over its 4 rows, across the 16 pixel lanes
// through the wave's (then free) residual rows in LDS, and across tiles with global_atomic_add_f64; pool_finish64 divides and
// rounds once.  Same bits as the two-per-CU kernel's integer sums; an image's mean does not depend on where the tile
// boundaries fall.
// KC = 2 (K = 64: res2's conv3): ring stages of 64 channels (128-byte rows), one stage per tile.
// DUAL (KC = 4): the 128-channel stage is [64 channels of x | 64 channels of x2] -- res2's first conv3 and its projection shortcut as
// one GEMM (`out += shortcut`, frcnn.py:970-977; the weights hold the concatenated rows); a DMA lane picks its source by chunk.
// STRIDED: 1x1 conv with a stride (the stride-2 projection shortcuts and first conv1s of res3 / res4): the DMA lane turns its output
// row into (image, ho, wo) and reads input pixel (ho * stride, wo * stride); everything behind the ring is unchanged.
template <int KC, int NW, int DBG = 0, bool POOL = false, bool DUAL = false, bool STRIDED = false>
__global__ __launch_bounds__(NW * 64, 1) void conv_ws_kernel(WsK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!DUAL || KC == 4, "two sources: 64 + 64 channels");
    constexpr int SCH = KC == 2 ? 64 : 128;  // channels per ring stage
    constexpr int KSL = SCH / 32;            // MFMA K steps per stage
    constexpr int PITCH = SCH * 2;           // bytes per pixel row of a stage
    constexpr int WS_SLOT = WS_BM * PITCH;
    constexpr int WS_NS = 6;                 // ring slots
    constexpr int WS_D = 5;                  // stages the DMA runs ahead
    constexpr int RING = WS_NS * WS_SLOT;
    constexpr int SPT = KC / KSL;            // ring stages per tile
    constexpr int NI = 16 / NW;              // 16-channel MFMA row tiles per wave
    constexpr int WCH = NI * 16;             // channels per wave
    constexpr int PPW = WS_SLOT / 1024 / NW; // DMA pieces per wave and ring stage
    constexpr int RPW = 32 / NW;             // residual DMA pieces per wave and tile
    constexpr int NU = NI * 2;               // epilogue units (32 channels x 16 pixels) per wave
    static_assert(!POOL || NW == 8, "the fused-mean epilogue is written for 32 channels per wave");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;

    // ---- which column block, which pixel tiles (see the header) ----
    const int wg = blockIdx.x, per_xcd = gridDim.x >> 3;
    const int xcd = wg & 7, i = wg >> 3;
    const int n_tile = i % p.n_tiles, lane_m = i / p.n_tiles;
    const int lanes_x = per_xcd / p.n_tiles;                 // M lanes per XCD
    const int ML = 8 * lanes_x;
    const int first = xcd * lanes_x + lane_m;
    const int ntw = first < p.m_tiles ? (p.m_tiles - first + ML - 1) / ML : 0;     // tiles of this workgroup
    const int n0 = n_tile * 256;
    const int total = ntw * SPT;                             // ring stages of this workgroup

    // ---- weights of the wave's 64 channels, every K step, in registers ----
    // row j of MFMA row tile ni is channel (ni>>1)*32 + (j>>2)*8 + (ni&1)*4 + (j&3) of the wave's 64: a lane ends up with 8
    // consecutive channels per pair of tiles = one 16-byte store, as in conv_mfma.hip
    half8 wf[NI][KC];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = n0 + wave * WCH + (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);
        const char *wr = p.w + (long)co * (DUAL ? 256 : p.kbytes) + g * 16;     // (DUAL: rows of 64 + 64 channels)
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) wf[ni][ks] = *reinterpret_cast<const half8 *>(wr + ks * 64);
    }
    char *res_lds = smem + RING + wave * (WS_BM * WCH * 2);       // this wave's 64 rows x WCH channels of residual
    float *bias_lds = reinterpret_cast<float *>(smem + RING + 4 * WS_BM * 128);
    if (tid < 256) bias_lds[tid] = p.bias[n0 + tid];
    ws_vm_wait<0>();                         // from here on vmcnt counts DMA pieces, residual loads and stores
    __syncthreads();

    // ---- LDS-DMA: piece q of the wave = rows (wave * PPW + q) * 4 + (lane >> 4) of the stage, 16-byte slot lane & 15 of the
    // 256-byte row, holding source chunk slot ^ (row & 15) (the swizzle sits on the source address) ----
    // (64-channel stages: 128-byte rows = whole cache lines, pieces of 8 rows, slot lane & 7 holding chunk slot ^ (row & 7))
    constexpr int LPR = PITCH / 16;          // lanes (16-byte slots) per row
    const int drow = lane / LPR, dslot = lane & (LPR - 1);
    auto request = [&](int q_, int qlo = 0, int qhi = WS_BM * (KC == 2 ? 128 : 256) / 1024 / NW) {   // q_ = global stage index of this workgroup; pieces [qlo, qhi = PPW)
        const int it = q_ / SPT, st = q_ - it * SPT;
        const int m0 = (first + it * ML) * WS_BM;
        char *dst = smem + (q_ % WS_NS) * WS_SLOT + wave * PPW * 1024;
#pragma unroll
        for (int q = qlo; q < qhi; ++q) {
            const int row = (wave * PPW + q) * (64 / LPR) + drow;
            const int m = min(m0 + row, p.M - 1);                             // rows past M are computed and dropped
            if constexpr (STRIDED) {
                const int n_ = m / p.HoWo, r_ = m - n_ * p.HoWo;
                const int ho = r_ / p.Wo, wo = r_ - ho * p.Wo;
                const long src = ((long)n_ * p.H + ho * p.stride) * p.W + wo * p.stride;
                VKW_GLDS16(p.x + src * p.kbytes + st * PITCH + ((dslot ^ (row & (LPR - 1))) << 4), dst + q * 1024);
            } else if constexpr (DUAL) {
                const int c = dslot ^ (row & (LPR - 1));                          // chunk 0-7: x, 8-15: x2 (one stage per tile)
                VKW_GLDS16((c < 8 ? p.x : p.x2) + (long)m * 128 + ((c & 7) << 4), dst + q * 1024);
            } else {
                VKW_GLDS16(p.x + (long)m * p.kbytes + st * PITCH + ((dslot ^ (row & (LPR - 1))) << 4), dst + q * 1024);
            }
        }
    };
    for (int q_ = 0; q_ < WS_D && q_ < total; ++q_) request(q_);

    // fragment address of pixel tile pt, step ksl of a stage: slot + pt * 4096 + row j * 256 + swizzled chunk
    int xoff[KSL];
#pragma unroll
    for (int ksl = 0; ksl < KSL; ++ksl) xoff[ksl] = j * PITCH + (((ksl * 4 + g) ^ (j & (LPR - 1))) << 4);

    // ---- main loop.  The epilogue is cut into 8 units (32 channels x 16 pixels).  (Measured on this kernel: 180 us of
    // epilogue + 320 us of MFMA + 190 us of loop skeleton add up to 698 us on the 512 -> 2048 layer at M = 200 704: with one
    // wave per SIMD the phases do not overlap.  A build with two accumulator sets that issued the previous tile's units
    // between the MFMA groups was correct and no faster (K = 256: 149 vs 136 us), so one set is kept.) ----
    unsigned long c_vm4[4] = {0, 0, 0, 0}, c_bar4[4] = {0, 0, 0, 0};
    unsigned long c_vm = 0, c_bar = 0, c_mma = 0, c_rw = 0, c_epi = 0, c_all = 0, t_a = 0, t_b = 0;   // DBG & 32
    if constexpr (DBG & 32) c_all = __builtin_amdgcn_s_memtime();
    floatx4 acc[4][NI];
    // 64-byte residual rows (NW = 8): chunk c of row r sits at slot c ^ RSW[(r >> 2) & 3], which spreads every 16-lane
    // group of the ds_read_b128 over all 64 banks
    auto rsw = [](int r) { return (0x1320 >> (((r >> 2) & 3) * 4)) & 3; };   // {0, 2, 3, 1}
    // The epilogue of a tile, specialised at compile time on residual / ReLU / "every row of the tile exists" so that it is one
    // basic block: all residual reads up front, packed fp32 adds, v_cvt_pk_f16_f32, ReLU on the rounded halves (the same
    // bits as max before rounding), unpredicated stores.  (The first version branched per unit on p.res / p.relu / m < M: every
    // unit paid a full LDS latency and ~60 VALU instructions -- 465 cycles per unit, 30 % of the kernel.)
    auto epilogue = [&](int it_, auto res_c, auto relu_c, auto full_c) {
        constexpr bool RES = decltype(res_c)::value, RELU = decltype(relu_c)::value, FULL = decltype(full_c)::value;
        const long mbase = (long)(first + it_ * ML) * WS_BM;
        // POOL: the tile's first image, the first row of its second one, and whether the tile reaches it (uniform)
        const long img0 = POOL ? mbase / p.HW : 0;
        const long m_split = (img0 + 1) * (POOL ? p.HW : 1);
        const bool two_images = POOL && m_split < mbase + WS_BM && m_split < p.M;
        unsigned *scr = reinterpret_cast<unsigned *>(res_lds);
        int ln = lane;                       // POOL: lane coordinates re-derived here, so that the addresses below are computed
        if constexpr (POOL) asm volatile("" : "+v"(ln));      // per tile instead of living in registers across the K loop
        const int gp = ln >> 4, jp = ln & 15;
#pragma unroll
        for (int u0 = 0; u0 < NU; u0 += 4) {                  // four units (32 channels x 64 pixels) at a time
            half8 rr[4];
            if constexpr (RES) {
#pragma unroll
                for (int u = u0; u < u0 + 4; ++u) {
                    const int qn = u >> 2, row = (u & 3) * 16 + j;
                    if constexpr (NW == 4)
                        rr[u - u0] = *reinterpret_cast<const half8 *>(res_lds + row * 128 + (((qn * 4 + g) ^ (row & 7)) << 4));
                    else
                        rr[u - u0] = *reinterpret_cast<const half8 *>(res_lds + row * 64 + ((g ^ rsw(row)) << 4));
                }
            }
#pragma unroll
            for (int u = u0; u < u0 + 4; ++u) {               // unit u = (qn, pt): 32 channels x 16 pixels
                const int qn = u >> 2, pt = u & 3;
                const int ch = wave * WCH + qn * 32 + g * 8;
                const floatx4 b0 = *reinterpret_cast<const floatx4 *>(bias_lds + ch), b1 = *reinterpret_cast<const floatx4 *>(bias_lds + ch + 4);
                floatx4 x0 = acc[pt][2 * qn] + b0, x1 = acc[pt][2 * qn + 1] + b1;
                if constexpr (RES) {
                    x0 += __builtin_convertvector(__builtin_shufflevector(rr[u - u0], rr[u - u0], 0, 1, 2, 3), floatx4);
                    x1 += __builtin_convertvector(__builtin_shufflevector(rr[u - u0], rr[u - u0], 4, 5, 6, 7), floatx4);
                }
                half4 h0 = __builtin_convertvector(x0, half4), h1 = __builtin_convertvector(x1, half4);
                half8 o = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                const long m = mbase + pt * 16 + j;
                if constexpr (POOL) {
                    // the lane's 4 channel pairs of row pt * 16 + j -> the wave's (consumed) residual rows, pair-major: dword
                    // [pair g * 4 + k][(row + g * 16) & 63]; the rotation spreads the 64 lanes over the 64 banks
                    if (!FULL && m >= p.M) o = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    const uintx4 ov = __builtin_bit_cast(uintx4, o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) scr[(gp * 4 + k) * 64 + ((pt * 16 + jp + gp * 16) & 63)] = ov[k];
                } else {
                    if constexpr (DBG & 64) {
                        asm volatile("" ::"v"(o));
                    } else {
                        if (FULL || m < p.M) *reinterpret_cast<half8 *>(p.y + (m * p.ldy + n0 + ch) * 2) = o;
                    }
                }
            }
        }
        if constexpr (POOL) {
            // lane (pair = lane & 15, quarter = lane >> 4) adds rows quarter * 16 ... + 15 of its two channels in fp64 (exact),
            // the four quarters meet through two lane exchanges, and lanes 0 ... 15 hand the tile's sums to the image's totals
            const int pr = jp, q = gp;
            const uintx4 *src = reinterpret_cast<const uintx4 *>(scr + pr * 64 + ((q + (pr >> 2)) & 3) * 16);
            uintx4 v[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) v[c4] = src[c4];
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            if (!two_images) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned d = v[r >> 2][r & 3];           // (a bit_cast straight from the vector element reads element 0)
                    const half2v h = __builtin_bit_cast(half2v, d);
                    a0 += (double)(float)h[0];
                    a1 += (double)(float)h[1];
                }
            } else {
                const int sloc = (int)(m_split - mbase);          // first row of the second image, 1 ... 63
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned d = v[r >> 2][r & 3];           // (a bit_cast straight from the vector element reads element 0)
                    const half2v h = __builtin_bit_cast(half2v, d);
                    const bool second = q * 16 + r >= sloc;
                    const float lo = (float)h[0], hi = (float)h[1];
                    a0 += (double)(second ? 0.f : lo);
                    a1 += (double)(second ? 0.f : hi);
                    b0 += (double)(second ? lo : 0.f);
                    b1 += (double)(second ? hi : 0.f);
                }
            }
            a0 += __shfl_xor(a0, 16);
            a1 += __shfl_xor(a1, 16);
            a0 += __shfl_xor(a0, 32);
            a1 += __shfl_xor(a1, 32);
            double *dst = p.pool + img0 * p.ldy + n0 + wave * WCH + pr * 2;
            if (ln < 16) {
                unsafeAtomicAdd(dst, a0);
                unsafeAtomicAdd(dst + 1, a1);
            }
            if (two_images) {
                b0 += __shfl_xor(b0, 16);
                b1 += __shfl_xor(b1, 16);
                b0 += __shfl_xor(b0, 32);
                b1 += __shfl_xor(b1, 32);
                if (ln < 16) {
                    unsafeAtomicAdd(dst + p.ldy, b0);
                    unsafeAtomicAdd(dst + p.ldy + 1, b1);
                }
            }
        }
    };
    for (int it = 0; it < ntw; ++it) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[pt][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
        // residual rows of this tile -> LDS: NW = 4: 8 pieces of 8 rows x 128 B of this wave's 64 channels, 16-byte slot
        // lane & 7 of a row holding chunk slot ^ (row & 7); NW = 8: 4 pieces of 16 rows x 64 B, slot lane & 3 holding chunk
        // slot ^ rsw(row)
        const int m0 = (first + it * ML) * WS_BM;
        if (p.res && !(DBG & 8)) {
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                if constexpr (NW == 4) {
                    const int row = q * 8 + (lane >> 3);
                    const int m = min(m0 + row, p.M - 1);
                    VKW_GLDS16(p.res + ((long)m * p.ldy + n0 + wave * 64) * 2 + (((lane & 7) ^ (row & 7)) << 4), res_lds + q * 1024);
                } else {
                    const int row = q * 16 + (lane >> 2);
                    const int m = min(m0 + row, p.M - 1);
                    VKW_GLDS16(p.res + ((long)m * p.ldy + n0 + wave * 32) * 2 + (((lane & 3) ^ rsw(row)) << 4), res_lds + q * 1024);
                }
            }
        }
#pragma unroll
        for (int st = 0; st < SPT; ++st) {
            const int gs = it * SPT + st;
            if constexpr (DBG & 32) t_a = __builtin_amdgcn_s_memtime();
            // stage gs has landed; the WS_D - 1 later stages stay in flight.  (Also letting the stores / residual requests issued
            // since stay in flight -- exact counts per stage of the tile -- was measured: no change, so the simple form stays.)
            if (gs + WS_D - 1 < total)
                ws_vm_wait<PPW * (WS_D - 1)>();
            else
                ws_vm_wait<0>();
            if constexpr (DBG & 32) {
                t_b = __builtin_amdgcn_s_memtime();
                c_vm += t_b - t_a;
                c_vm4[st & 3] += t_b - t_a;
            }
            __builtin_amdgcn_s_barrier();                    // ... for every wave; and stage gs - 1 has been read by all
            if constexpr (DBG & 32) {
                t_a = __builtin_amdgcn_s_memtime();
                c_bar += t_a - t_b;
                c_bar4[st & 3] += t_a - t_b;
            }
            const bool more = gs + WS_D < total && !(DBG & 4);   // a stage to request into the slot stage gs - 1 just left
            const char *slot = smem + (gs % WS_NS) * WS_SLOT;
            // all 16 fragments of the stage are requested up front; the four DMA pieces of the stage WS_D ahead are issued
            // BETWEEN the MFMA groups, where their issue cost (~100 cycles each) hides under the matrix pipe
            half8 xf[KSL][4];
#pragma unroll
            for (int ksl = 0; ksl < KSL; ++ksl)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) xf[ksl][pt] = *reinterpret_cast<const half8 *>(slot + pt * 16 * PITCH + xoff[ksl]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ksl = 0; ksl < KSL; ++ksl) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if constexpr (DBG & 2) {
                            asm volatile("" ::"v"(xf[ksl][pt]));
                            continue;
                        }
                        acc[pt][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][KSL * st + ksl], xf[ksl][pt], acc[pt][ni], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NW == 4) {
                    if (more) request(gs + WS_D, ksl, ksl + 1);
                } else {
                    if (more && (ksl & 1)) request(gs + WS_D, ksl >> 1, (ksl >> 1) + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DBG & 32) c_mma += __builtin_amdgcn_s_memtime() - t_a;
        }
        // ---- epilogue: (acc + bias) + residual, ReLU, f16.  The residual pieces are older than the SPT stages requested since ----
        if constexpr (DBG & 32) t_a = __builtin_amdgcn_s_memtime();
        if constexpr (DBG & 1) {                 // (keep the MFMAs alive in the timing-only build without an epilogue)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(acc[pt][ni]));
        }
        if (!(DBG & 1)) {
            if (p.res) {
                if ((it + 1) * SPT - 1 + WS_D < total)
                    ws_vm_wait<PPW * SPT>();
                else
                    ws_vm_wait<0>();
            }
            if constexpr (DBG & 32) {
                t_b = __builtin_amdgcn_s_memtime();
                c_rw += t_b - t_a;
            }
            const bool full = (long)(first + it * ML + 1) * WS_BM <= p.M;
            auto by_full = [&](auto r, auto l) {
                if (full)
                    epilogue(it, r, l, std::true_type{});
                else
                    epilogue(it, r, l, std::false_type{});
            };
            auto by_relu = [&](auto r) {
                if (p.relu)
                    by_full(r, std::true_type{});
                else
                    by_full(r, std::false_type{});
            };
            if (p.res)
                by_relu(std::true_type{});
            else
                by_relu(std::false_type{});
            if constexpr (DBG & 32) c_epi += __builtin_amdgcn_s_memtime() - t_b;
        }
    }
    if constexpr (DBG & 32) {
        if (lane == 0) {
            unsigned long *o = p.stamps + ((long)blockIdx.x * NW + wave) * 16;
            o[0] = __builtin_amdgcn_s_memtime() - c_all;
            o[1] = c_vm;
            o[2] = c_bar;
            o[3] = c_mma;
            o[4] = c_rw;
            o[5] = c_epi;
            o[6] = ntw;
            o[7] = __builtin_amdgcn_s_memrealtime();
            for (int q = 0; q < 4; ++q) {
                o[8 + q] = c_vm4[q];
                o[12 + q] = c_bar4[q];
            }
        }
    }
}

// The fused-mean form (a.pool_part set) runs here when K = 512 and the column blocks tile an XCD; otherwise on the two-per-CU
// kernel.  The per-image fp64 sums (Cout x 8 B per image) live in the workspace that kernel sizes for its per-tile partials
// (Cout x 16 B per 128 rows: larger from 128 rows per image on, which its own eligibility asks for).  Both sides of the
// hand-over -- launch_conv and launch_pool_finish -- decide with this one function.
bool conv_pool_sums_f64(long N, int HW, int Cin, int Cout, bool dual) {
    const char *v = getenv("VK_CONV_WS");                // "0" disables (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return false;
    const char *q = getenv("VK_WS_POOL");                // "0": the fused mean stays on the two-per-CU kernel
    if (q && q[0] == '0') return false;
    if (dual || Cin != 512 || Cout % 256 != 0 || 32 % (Cout / 256) != 0) return false;
    const long M = N * HW;
    return HW >= 128 && HW <= 8192 && M >= 8 * 128 && M < (1L << 31) - 128;
}

bool conv_ws_pool_ok(const ConvArgs &a) {
    return a.pool_part && conv_duo_pool_ok(a) && a.relu <= 1 && a.ldy == a.Cout &&
           conv_pool_sums_f64(a.N, a.Ho * a.Wo, a.Cin, a.Cout, a.x2 != nullptr);
}

bool conv_ws_eligible(const ConvArgs &a) {
    const char *v = getenv("VK_CONV_WS");                // "0" disables (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return false;
    if (a.pool_part) return conv_ws_pool_ok(a);
    if (a.stem || a.groups > 1 || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return false;
    if (a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride < 1) return false;
    if (a.stride != 1) {                                                  // strided: K = 256 / 512, one source, no residual rows to map
        const char *sv = getenv("VK_WS_STRIDED");                         // "0": strided 1x1 convs stay on the two-per-CU kernel (A/B switch)
        if ((sv && sv[0] == '0') || a.x2 || a.res || (a.Cin != 256 && a.Cin != 512)) return false;
    }
    if (a.x2 && (a.Cin != 64 || a.Cin2 != 64)) return false;             // two sources: 64 + 64 channels only
    if (a.Cout % 256 != 0 || a.ldy != a.Cout || (a.Cin != 64 && a.Cin != 128 && a.Cin != 256 && a.Cin != 512)) return false;
    const int nt = a.Cout / 256;
    if (32 % nt != 0) return false;                      // column blocks must tile the 32 workgroups of an XCD
    const long M = (long)a.N * a.Ho * a.Wo;
    return M >= 8 * 128 && M < (1L << 31) - 128;      // (a.Cin % 128 == 0: whole 128-channel ring stages)
}

template <int KC, int NW = 8, int DBG = 0, bool POOL = false, bool DUAL = false, bool STRIDED = false>
static int launch_ws(const WsK &k, hipStream_t stream) {
    constexpr int smem = 6 * WS_BM * 256 + 4 * WS_BM * 128 + 1024;
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_kernel<KC, NW, DBG, POOL, DUAL, STRIDED>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_ws_kernel<KC, NW, DBG, POOL, DUAL, STRIDED>), dim3(256), dim3(NW * 64), smem, stream, k);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_conv_ws(const ConvArgs &a, hipStream_t stream) {
    WsK k;
    k.x = (const char *)a.x;
    k.x2 = (const char *)a.x2;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.res = (const char *)a.res;
    k.y = (char *)a.y;
    const long M = (long)a.N * a.Ho * a.Wo;
    k.M = (int)M;
    k.kbytes = a.Cin * 2;
    k.ldy = a.ldy;
    k.relu = a.relu;
    k.m_tiles = (int)((M + WS_BM - 1) / WS_BM);
    k.n_tiles = a.Cout / 256;
    k.stamps = nullptr;
    k.pool = (double *)a.pool_part;
    k.HW = a.Ho * a.Wo;
    k.H = a.H;
    k.W = a.W;
    k.Wo = a.Wo;
    k.HoWo = a.Ho * a.Wo;
    k.stride = a.stride;

    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    int st;
    if (a.pool_part) {                                   // fused mean: zero the per-image sums, then the POOL build
        VK_CHECK_HIP(hipMemsetAsync(k.pool, 0, (size_t)(M / k.HW) * a.Cout * sizeof(double), stream));
        VK_TRY((launch_ws<16, 8, 0, true>(k, stream)));
        if (tm) {
            VK_CHECK_HIP(hipEventRecord(e1, stream));
            tm->recs.push_back({a.concurrent ? 6 : 8, 2.0 * (double)M * a.Cout * a.Cin, e0, e1, (int)M, a.Cout, a.Cin, 1, 1,
                                2.0 * ((double)M * a.Cin + (double)M * a.Cout * (a.res ? 1 : 0) + (double)a.Cout * a.Cin)});
        }
        return VK_OK;
    }
    const char *nwv = getenv("VK_WS_WAVES");             // "4" / "8": A/B switch, re-read per call
    const int nw = nwv ? atoi(nwv) : 8;
#ifdef VK_ABLATION      // stamp build: tools/ builds only (make ABLATION=1)
    if (const char *sf = getenv("VK_WS_STAMPS")) {       // diagnostic: one stamped launch (K = 512), cycle sums appended to the file
        if (a.Cin != 512) return VK_EINVAL;
        const size_t nb = (size_t)256 * 8 * 16 * sizeof(unsigned long);
        VK_CHECK_HIP(hipMalloc((void **)&k.stamps, nb));
        VK_CHECK_HIP(hipMemsetAsync(k.stamps, 0, nb, stream));
        st = nw == 8 ? launch_ws<16, 8, 32>(k, stream) : launch_ws<16, 4, 32>(k, stream);
        VK_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long> h(256 * 8 * 16);
        VK_CHECK_HIP(hipMemcpy(h.data(), k.stamps, nb, hipMemcpyDeviceToHost));
        VK_CHECK_HIP(hipFree(k.stamps));
        if (FILE *f = fopen(sf, "a")) {
            fprintf(f, "# wg wave total vm_wait barrier stage_body res_wait epilogue tiles realtime\n");
            for (int w = 0; w < 256 * nw; ++w) {
                fprintf(f, "%d %d", w / nw, w % nw);
                for (int i = 0; i < 16; ++i) fprintf(f, " %lu", h[(size_t)w * 16 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    } else
#endif
    {
#ifdef VK_ABLATION
        const char *d = getenv("VK_WS_DBG");
        const int dbg = d ? atoi(d) : 0;
#define VKW_DBG_CASE(NW_, D_) \
    case D_: st = launch_ws<16, NW_, D_>(k, stream); break;
#else      // the shipped library has no timing-only (WRONG-result) builds and ignores VK_WS_DBG / VK_WS_STAMPS
        const int dbg = 0;
#define VKW_DBG_CASE(NW_, D_)
#endif
        if (a.stride != 1)
            st = a.Cin == 256 ? launch_ws<8, 8, 0, false, false, true>(k, stream) : launch_ws<16, 8, 0, false, false, true>(k, stream);
        else if (a.x2)
            st = launch_ws<4, 8, 0, false, true>(k, stream);
        else if (a.Cin == 64)
            st = nw == 8 ? launch_ws<2, 8>(k, stream) : launch_ws<2, 4>(k, stream);
        else if (a.Cin == 128)
            st = nw == 8 ? launch_ws<4, 8>(k, stream) : launch_ws<4, 4>(k, stream);
        else if (a.Cin == 256)
            st = nw == 8 ? launch_ws<8, 8>(k, stream) : launch_ws<8, 4>(k, stream);
        else if (nw == 8) {
            switch (dbg) {
                VKW_DBG_CASE(8, 1) VKW_DBG_CASE(8, 2) VKW_DBG_CASE(8, 3) VKW_DBG_CASE(8, 4) VKW_DBG_CASE(8, 8) VKW_DBG_CASE(8, 12)
                VKW_DBG_CASE(8, 15) VKW_DBG_CASE(8, 64) VKW_DBG_CASE(8, 9) VKW_DBG_CASE(8, 72)
                default: st = launch_ws<16, 8>(k, stream); break;
            }
        } else {
            switch (dbg) {
                VKW_DBG_CASE(4, 1) VKW_DBG_CASE(4, 2) VKW_DBG_CASE(4, 3) VKW_DBG_CASE(4, 4) VKW_DBG_CASE(4, 8) VKW_DBG_CASE(4, 12)
                VKW_DBG_CASE(4, 15)
                default: st = launch_ws<16, 4>(k, stream); break;
            }
        }
#undef VKW_DBG_CASE
    }
    VK_TRY(st);
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const int K = a.Cin + (a.x2 ? a.Cin2 : 0);
        tm->recs.push_back({a.concurrent ? 6 : 8, 2.0 * (double)M * a.Cout * K, e0, e1, (int)M, a.Cout, K, 1, 1,
                            2.0 * ((double)M * K + (double)M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * K)});
    }
    return VK_OK;
}

}  // namespace vk
