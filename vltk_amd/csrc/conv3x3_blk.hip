// 3x3 convolution (stride 1, pad == dilation) for NARROW channel blocks: the grouped 3x3 of ResNeXt bottlenecks
// (BottleneckBlock conv2 with groups = NUM_GROUPS, reference vltk/modeling/frcnn.py:942-952: 8 / 16 / 32 / 64 channels per
// group) and the dense 64 -> 64 conv2 of res2 (:934-952 with groups 1).
//
// Why a kernel of its own: these layers carry little arithmetic per byte (a 32x8d res2 conv2 is 37 kFLOP per pixel
// against 1 KB of input + output), so they are bound by memory, not by the matrix cores.  The im2col kernel they used to
// run on (conv_mfma.hip) re-fetches every input pixel once per tap AND once per 64-channel output tile, and multiplies
// the zero blocks of the slice-diagonal weights (MFMA efficiency = channels per group / 64).  Here
//   * the unit of work is a SLAB of 64 channels (= one 128-byte line per pixel: one dense 64-channel block, or two
//     blocks of 32 channels that hold 4 / 2 / 1 whole groups) of a 2-D spatial tile: the tile's pixels plus halo come
//     into LDS ONCE (zero-filled outside the image) and all nine taps read them there;
//   * the slab's weights live in REGISTERS for the lifetime of the workgroup (36 fragments for a 64-wide block, 18 for a
//     32-wide one), and a workgroup walks many tiles of its slab, so weights cost no memory traffic in steady state;
//   * only the MFMAs of the diagonal blocks are issued (32-wide blocks: half of them), the products with the
//     structurally-zero weights are skipped -- they are exact zeros, so the result is bit-identical to the im2col kernel's
//     (same K order: tap-major, 32 channels per step);
//   * one 512-thread workgroup per CU; the NEXT tile arrives by LDS-DMA in a second LDS buffer while this one is multiplied.
// Tile shapes: 8 x 32 output pixels (dilation 1, any image size) for the backbone; the whole 14 x 14 RoI with dilation 2
// for the Res5 head (:1344-1355), run as rows of 16 with 14 valid columns.  LDS image: [padded pixel][8 x 16-B chunks],
// chunk' = chunk ^ (padded column & 7): conflict-free ds_read_b128 for the 16x16x32 operand lane map.
// Weights are read in the layout vk_pack_conv_weight already produces ([cout][9 taps][64-channel slice], zero outside a
// channel's group), so nothing about the packing or the ABI changes.
#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct BlkK {
    const char *x;
    const char *w;
    const float *bias;
    char *y;
    int N, H, W;
    int cbytes;              // C * 2
    int wrow_bytes;          // 9 * 64 * 2
    int relu;
    int tiles_x, tiles_y, ntiles;
    const char *zero;        // >= 16 zero bytes: source of the out-of-image pixels
};

#define VKB_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void blk_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One 512-thread workgroup per CU (two waves per SIMD).  The tile of step t+1 is brought into the other LDS buffer by
// LDS-DMA (no registers) while the waves multiply tile t: per wave NQ DMA instructions of 1 KiB = 8 padded pixels x 128 B,
// the XOR swizzle applied on the per-lane SOURCE address (the LDS image of a DMA is lane-linear), out-of-image pixels
// fetched from a zero page.  One counted vmcnt + one raw barrier hands a buffer over; a second barrier frees it.
// TW: tile width, a multiple of 16 (a 16-pixel MFMA column block never wraps a tile row, so a lane's LDS addresses are
// `per-lane constant + per-block scalar`); VW <= TW: columns that exist (the 14 x 14 RoI runs as TW = 16, VW = 14).
template <int CB, int TH, int TW, int VW, int DIL>
__global__ __launch_bounds__(512, 2) void conv3x3_blk_kernel(BlkK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(TW % 16 == 0 && VW <= TW, "tile width");
    constexpr int PH = TH + 2 * DIL, PW = TW + 2 * DIL;
    constexpr int CBL = TW / 16;                // column blocks per tile row
    constexpr int NPB = TH * CBL;               // 16-pixel blocks per tile
    constexpr int KS = CB / 32;                 // 32-channel MFMA steps per tap
    constexpr int NQ = ((PH * PW + 7) / 8 + 7) / 8;   // DMA instructions per wave and tile
    constexpr int BUF = NQ * 8 * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cp = wave & 1;                    // which 32 output channels of the slab this wave produces
    const int par = wave >> 1;                  // which pixel blocks: par, par + 4, ...
    const int g = lane >> 4, j = lane & 15;
    const int slab = blockIdx.y;
    const int cbase = (CB == 32) ? cp * 32 : 0; // first input channel (within the slab) of this wave's block

    // ---- the wave's weights: 2 MFMA row tiles x 9 taps x KS steps, resident in registers ----
    // row j of tile ni is output channel (j>>2)*8 + ni*4 + (j&3) of the wave's 32, so that a lane ends up with 8
    // consecutive channels (one 16-byte store), as in conv_mfma.hip
    half8 wf[2][9][KS];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int co = slab * 64 + cp * 32 + (j >> 2) * 8 + ni * 4 + (j & 3);
        const char *wr = p.w + (long)co * p.wrow_bytes + (cbase + g * 8) * 2;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[ni][tap][ks] = *reinterpret_cast<const half8 *>(wr + (tap * 64 + ks * 32) * 2);
    }
    // the slab's 64 biases sit behind the two tile buffers (read back per block: 8 VGPRs the fragment sets need)
    float *bias_lds = reinterpret_cast<float *>(smem + 2 * BUF);
    if (tid < 64) bias_lds[tid] = p.bias[slab * 64 + tid];
    const float *bl = bias_lds + cp * 32 + g * 8;
    // LDS byte offset of this lane's fragment for tap column dx and step ks, relative to the block's first pixel of
    // the tap row: pixel (j + dx * DIL) * 128 B + swizzled 16-B chunk (the key is the padded column mod 8; 16-pixel
    // blocks keep it independent of the block)
    // (step ks = 1 reads chunk + 4: the same offset with bit 6 flipped, all bases being multiples of 128)
    int colb[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pc = j + dx * DIL;
        colb[dx] = pc * 128 + (((cbase / 8 + g) ^ (pc & 7)) << 4);
    }
    blk_vm_wait<0>();                           // the weights are in: from here on vmcnt counts DMA pieces and stores only
    __syncthreads();                            // ... and the biases are in LDS

    // DMA geometry of this lane: instruction q of the wave covers padded pixels (wave * NQ + q) * 8 + (lane >> 3).  What
    // does not depend on the tile is computed once (the padded (row, column) of the first piece; the others follow by
    // adding 8 columns with at most two row wraps); offsets inside an image are 32-bit.
    const int tpi = p.tiles_x * p.tiles_y;
    const int P0 = wave * NQ * 8 + (lane >> 3);
    const int pr0 = P0 / PW, pc0 = P0 - pr0 * PW;   // piece q: 8 q pixels further along the padded rows
    auto request = [&](int t, int buf) {
        const int n = t / tpi;
        const int tr = t - n * tpi;
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int y0 = ty * TH - DIL, x0 = tx * VW - DIL;
        // first padded pixel of the tile (may lie outside the image: only in-image lanes dereference it)
        const char *org = p.x + ((long)n * p.H * p.W + (long)y0 * p.W + x0) * p.cbytes + slab * 128;
        const unsigned nrow = (unsigned)p.H, ncol = (unsigned)p.W;
#pragma clang loop unroll(disable)
        for (int q = 0; q < NQ; ++q) {
            static_assert(8 * (NQ - 1) <= 2 * PW, "a piece is at most two row wraps from the first");
            int pr = pr0, pc = pc0 + 8 * q;
            if (pc >= PW) pc -= PW, ++pr;
            if (pc >= PW) pc -= PW, ++pr;
            const bool ok = P0 + 8 * q < PH * PW && (unsigned)(y0 + pr) < nrow && (unsigned)(x0 + pc) < ncol;
            const char *src = ok ? org + ((pr * p.W + pc) * p.cbytes + (((lane & 7) ^ (pc & 7)) << 4)) : p.zero;
            VKB_GLDS16(src, smem + buf * BUF + (wave * NQ + q) * 1024);
            __builtin_amdgcn_sched_barrier(0);          // one piece's address at a time (registers)
        }
    };

    int t = blockIdx.x;
    if (t < p.ntiles) request(t, 0);
    int it = 0;
    for (; t < p.ntiles; t += gridDim.x, ++it) {
        const int tn = t + gridDim.x;
        const char *tile = smem + (it & 1) * BUF;
        if (tn < p.ntiles) {
            request(tn, (it + 1) & 1);
            blk_vm_wait<NQ>();                  // all but the NQ pieces just issued: tile t has landed, older stores are done
        } else {
            blk_vm_wait<0>();
        }
        __builtin_amdgcn_s_barrier();           // every wave's pieces of tile t are in LDS

        const int n = t / tpi;
        const int tr = t - n * tpi;
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int y0 = ty * TH, x0 = tx * VW;
        // ---- 16 pixels x 32 channels per block: 9 taps x KS steps, two row tiles.  A wave's blocks are pb = par + 4 i.
        // Fragment reads run one tap row (3 * KS fragments) AHEAD of the MFMAs in two alternating register sets, across
        // block boundaries too, so the LDS latency sits under the previous row's MFMAs ----
        half8 xa[3 * KS], xb[3 * KS];
        floatx4 acc0, acc1;
        // block i exists for every wave when 4 i + 3 < NPB (a compile-time fact: no branch, so hipcc keeps COUNTED lgkmcnt
        // waits across the straight-line code), otherwise only for the waves with par + 4 i < NPB
        auto exists = [&](int i) { return 4 * i + 3 < NPB || par + 4 * i < NPB; };
        auto rd = [&](half8 (&x)[3 * KS], int i, int dy) {
            const int pb = par + 4 * i;
            if (4 * i >= NPB) return;
            if (!exists(i)) return;                                      // wave-uniform
            const int r = pb / CBL, cb = pb - r * CBL;
            const char *rowp = tile + (r * PW + cb * 16) * 128 + dy * DIL * PW * 128;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) x[dx * KS + ks] = *reinterpret_cast<const half8 *>(rowp + (colb[dx] ^ (ks * 64)));
        };
        auto mm = [&](const half8 (&x)[3 * KS], int dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][dy * 3 + dx][ks], x[dx * KS + ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][dy * 3 + dx][ks], x[dx * KS + ks], acc1, 0, 0, 0);
                }
        };
        auto out = [&](int i) {
            const int pb = par + 4 * i;
            const int r = pb / CBL, cb = pb - r * CBL;
            const int c = cb * 16 + j;
            const int oy = y0 + r, ox = x0 + c;
            if (c < VW && oy < p.H && ox < p.W) {
                half8 o;
                const floatx4 b0 = *reinterpret_cast<const floatx4 *>(bl), b1 = *reinterpret_cast<const floatx4 *>(bl + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0 = acc0[e] + b0[e], a1 = acc1[e] + b1[e];
                    if (p.relu) {
                        a0 = a0 > 0.f ? a0 : 0.f;
                        a1 = a1 > 0.f ? a1 : 0.f;
                    }
                    o[e] = (_Float16)a0;
                    o[4 + e] = (_Float16)a1;
                }
                *reinterpret_cast<half8 *>(p.y + (((long)n * p.H + oy) * p.W + ox) * p.cbytes + slab * 128 + (cp * 32 + g * 8) * 2) = o;
            }
        };
        // block i with row 0 already in X: rows 1, 2 and the next block's row 0 are requested before rows 0, 1, 2 are multiplied
#define VKB_BLOCK(X, Y, i)                                  \
    if (exists(i)) {                                        \
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
        static_assert(NPB <= 16, "a wave walks at most four pixel blocks per tile");
        rd(xa, 0, 0);
        VKB_BLOCK(xa, xb, 0)
        VKB_BLOCK(xb, xa, 1)
        VKB_BLOCK(xa, xb, 2)
        VKB_BLOCK(xb, xa, 3)
#undef VKB_BLOCK
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // every wave is done reading this buffer: the step after next may refill it
    }
}

// what the kernel takes (see the header): everything else stays on conv_mfma.hip
static int blk_variant(const ConvArgs &a) {
    const char *v = getenv("VK_CONV3X3_BLK");            // "0" disables (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return 0;
    if (a.stem || a.x2 || a.pool_part || a.res || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return 0;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != a.dil || a.Cin != a.Cout || a.ldy != a.Cout) return 0;
    if (a.Cin % 64 != 0 || a.H != a.Ho || a.W != a.Wo) return 0;
    if ((long)(a.H + 8) * a.W * a.Cin * 2 >= (1L << 31)) return 0;      // 32-bit in-image offsets: larger images stay on conv_mfma.hip
    int cb;
    if (a.groups > 1) {
        if (a.Cin % a.groups != 0) return 0;
        const int cg = a.Cin / a.groups;
        if (cg != 8 && cg != 16 && cg != 32 && cg != 64) return 0;
        cb = cg <= 32 ? 32 : 64;
    } else {
        if (a.Cin != 64) return 0;
        cb = 64;
    }
    if (a.dil == 1) return cb == 32 ? 1 : 2;
    if (a.dil == 2 && a.H == 14 && a.W == 14) return cb == 32 ? 3 : 4;
    return 0;
}

bool conv3x3_blk_eligible(const ConvArgs &a) { return blk_variant(a) != 0; }

template <int CB, int TH, int TW, int VW, int DIL>
static int launch_blk(const ConvArgs &a, hipStream_t stream) {
    constexpr int smem = 2 * ((((TH + 2 * DIL) * (TW + 2 * DIL) + 7) / 8 + 7) / 8) * 8 * 1024 + 256;
    // per device: the zero page lives in the memory of the device that reads it, and the LDS attribute is set on each
    // device's copy of the code object (a process may hold handles on several GPUs: vk_create takes a device index)
    int dev = 0;
    VK_CHECK_HIP(hipGetDevice(&dev));
    VK_REQUIRE(dev >= 0 && dev < VK_MAX_DEVICES, VK_EINVAL, "conv3x3_blk: device index %d", dev);
    static char *zero_pages[VK_MAX_DEVICES] = {};
    if (!zero_pages[dev]) {
        VK_CHECK_HIP(hipMalloc((void **)&zero_pages[dev], 256));
        VK_CHECK_HIP(hipMemset(zero_pages[dev], 0, 256));
    }
    char *zero_page = zero_pages[dev];
    static bool attr_set[VK_MAX_DEVICES] = {};
    if (!attr_set[dev]) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_blk_kernel<CB, TH, TW, VW, DIL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set[dev] = true;
    }
    BlkK k;
    k.x = (const char *)a.x;
    k.w = (const char *)a.w;
    k.bias = a.bias;
    k.y = (char *)a.y;
    k.N = a.N;
    k.H = a.H;
    k.W = a.W;
    k.cbytes = a.Cin * 2;
    k.wrow_bytes = 9 * 64 * 2;
    k.relu = a.relu;
    k.tiles_x = ceil_div(a.W, VW);
    k.tiles_y = ceil_div(a.H, TH);
    const long nt = (long)a.N * k.tiles_x * k.tiles_y;
    VK_REQUIRE(nt > 0 && nt < (1L << 31), VK_EINVAL, "conv3x3_blk: %ld tiles", nt);
    k.ntiles = (int)nt;
    k.zero = zero_page;
    const int slabs = a.Cin / 64;
    // one workgroup per CU over all slabs; a workgroup keeps its slab's weights in registers across its tiles
    int gx = 256 / slabs;
    if (gx < 1) gx = 1;
    if (gx > k.ntiles) gx = k.ntiles;
    const int cg = a.groups > 1 ? a.Cin / a.groups : a.Cin;
    const long M = (long)a.N * a.H * a.W;
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL((conv3x3_blk_kernel<CB, TH, TW, VW, DIL>), dim3(gx, slabs), dim3(512), smem, stream, k);
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        tm->recs.push_back({a.concurrent ? 6 : 7, 2.0 * (double)M * a.Cout * 9.0 * cg, e0, e1, (int)M, a.Cout, a.Cin, 3, 1,
                            2.0 * (double)M * a.Cin * 2.0 + (double)a.Cout * 9.0 * cg * 2.0});
    }
    return VK_OK;
}

int launch_conv3x3_blk(const ConvArgs &a, hipStream_t stream) {
    switch (blk_variant(a)) {
        case 1: return launch_blk<32, 8, 32, 32, 1>(a, stream);
        case 2: return launch_blk<64, 8, 32, 32, 1>(a, stream);
        case 3: return launch_blk<32, 14, 16, 14, 2>(a, stream);
        case 4: return launch_blk<64, 14, 16, 14, 2>(a, stream);
    }
    set_error("conv3x3_blk: shape not eligible");
    return VK_EINVAL;
}

}  // namespace vk
