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
//   * two workgroups per CU (<= 44 KiB LDS, <= 256 VGPRs): one loads its next tile while the other multiplies.
// Tile shapes: 8 x 32 output pixels (dilation 1, any image size) for the backbone; the whole 14 x 14 RoI with dilation 2
// for the Res5 head (:1344-1355).  LDS image: [padded pixel][8 x 16-B chunks], chunk' = chunk ^ key with
// key = (col + (TW mod 8) * row) & 7: conflict-free ds_read_b128 for the 16x16x32 operand lane map.
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
};

template <int CB, int TH, int TW, int DIL>
__global__ __launch_bounds__(256, 2) void conv3x3_blk_kernel(BlkK p) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    constexpr int PH = TH + 2 * DIL, PW = TW + 2 * DIL;
    constexpr int NPX = TH * TW, NPB = (NPX + 15) / 16;
    constexpr int KS = CB / 32;                 // 32-channel MFMA steps per tap
    constexpr int KEYM = TW & 7;
    constexpr int NLD = (PH * PW * 8 + 255) / 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cp = wave & 1;                    // which 32 output channels of the slab this wave produces
    const int par = wave >> 1;                  // which pixel blocks (even / odd)
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
    float b[8];
    {
        const float *bp = p.bias + slab * 64 + cp * 32 + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = bp[e];
    }
    const int chunk0 = cbase / 8 + g;           // + ks * 4

    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int n = t / (p.tiles_x * p.tiles_y);
        const int tr = t - n * (p.tiles_x * p.tiles_y);
        const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        const char *img = p.x + (long)n * p.H * p.W * p.cbytes + slab * 128;

        // ---- tile + halo -> LDS (zero outside the image) ----
        u32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int P = idx >> 3, c = idx & 7;
            const int pr = P / PW, pc = P - pr * PW;
            const int yy = y0 - DIL + pr, xx = x0 - DIL + pc;
            const bool ok = idx < PH * PW * 8 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            v[i] = ok ? *reinterpret_cast<const u32x4 *>(img + ((long)yy * p.W + xx) * p.cbytes + c * 16) : u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();                        // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * 256;
            const int P = idx >> 3, c = idx & 7;
            const int pr = P / PW, pc = P - pr * PW;
            if (idx < PH * PW * 8) *reinterpret_cast<u32x4 *>(tile + P * 128 + ((c ^ ((pc + KEYM * pr) & 7)) << 4)) = v[i];
        }
        __syncthreads();

        // ---- 16 pixels x 32 channels per step: 9 taps x KS steps, two row tiles ----
        for (int pb = par; pb < NPB; pb += 2) {
            const int pi = pb * 16 + j;
            const int pv = pi < NPX ? pi : NPX - 1;
            const int r = pv / TW, c = pv - r * TW;
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int pr = r + (tap / 3) * DIL, pc = c + (tap % 3) * DIL;
                const char *src = tile + (pr * PW + pc) * 128;
                const int key = (pc + KEYM * pr) & 7;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 xf = *reinterpret_cast<const half8 *>(src + (((chunk0 + ks * 4) ^ key) << 4));
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][tap][ks], xf, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][tap][ks], xf, acc1, 0, 0, 0);
                }
            }
            const int oy = y0 + r, ox = x0 + c;
            if (pi < NPX && oy < p.H && ox < p.W) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0 = acc0[e] + b[e], a1 = acc1[e] + b[4 + e];
                    if (p.relu) {
                        a0 = a0 > 0.f ? a0 : 0.f;
                        a1 = a1 > 0.f ? a1 : 0.f;
                    }
                    o[e] = (_Float16)a0;
                    o[4 + e] = (_Float16)a1;
                }
                *reinterpret_cast<half8 *>(p.y + (((long)n * p.H + oy) * p.W + ox) * p.cbytes + slab * 128 + (cp * 32 + g * 8) * 2) = o;
            }
        }
    }
}

// what the kernel takes (see the header): everything else stays on conv_mfma.hip
static int blk_variant(const ConvArgs &a) {
    const char *v = getenv("VK_CONV3X3_BLK");            // "0" disables (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return 0;
    if (a.stem || a.x2 || a.pool_part || a.res || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return 0;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != a.dil || a.Cin != a.Cout || a.ldy != a.Cout) return 0;
    if (a.Cin % 64 != 0 || a.H != a.Ho || a.W != a.Wo) return 0;
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

template <int CB, int TH, int TW, int DIL>
static int launch_blk(const ConvArgs &a, hipStream_t stream) {
    constexpr int smem = (TH + 2 * DIL) * (TW + 2 * DIL) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_blk_kernel<CB, TH, TW, DIL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
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
    k.tiles_x = ceil_div(a.W, TW);
    k.tiles_y = ceil_div(a.H, TH);
    const long nt = (long)a.N * k.tiles_x * k.tiles_y;
    VK_REQUIRE(nt > 0 && nt < (1L << 31), VK_EINVAL, "conv3x3_blk: %ld tiles", nt);
    k.ntiles = (int)nt;
    const int slabs = a.Cin / 64;
    // two workgroups per CU over all slabs; a workgroup keeps its slab's weights in registers across its tiles
    int gx = 512 / slabs;
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
    hipLaunchKernelGGL((conv3x3_blk_kernel<CB, TH, TW, DIL>), dim3(gx, slabs), dim3(256), smem, stream, k);
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
        case 1: return launch_blk<32, 8, 32, 1>(a, stream);
        case 2: return launch_blk<64, 8, 32, 1>(a, stream);
        case 3: return launch_blk<32, 14, 14, 2>(a, stream);
        case 4: return launch_blk<64, 14, 14, 2>(a, stream);
    }
    set_error("conv3x3_blk: shape not eligible");
    return VK_EINVAL;
}

}  // namespace vk
