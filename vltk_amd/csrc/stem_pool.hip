// BasicStem as ONE kernel (f16): 7x7 stride-2 conv + folded BN + ReLU + 3x3 stride-2 max-pool
// (reference vltk/modeling/frcnn.py:872-879: F.relu_(conv1(x)); max_pool2d 3/2 -- pad 0 + ceil_mode in the caffe-style
// checkpoints, pad 1 otherwise).
//
// Why: run as two kernels the stem writes its 64-channel map at half resolution (1.09 GB at 32 x 800 x 1333) only for the
// pool to read it back and keep a quarter of it; on the generic im2col kernel the conv itself ran at 237 TFLOP/s.  Here the
// conv output of a pool tile lives in LDS only.
//
//   * unit of work: 4 x 16 pooled pixels of one image = 9 x 33 stem pixels (the windows overlap by one) = 23 x 72 pixels of the
//     zero-bordered NHWC4 image (stem_pack's output: 8 B per pixel), staged in LDS once (13 KiB);
//   * the conv is an MFMA GEMM with K = 7 kernel rows x 32 (7 taps x 4 channels + 4 zero-weighted elements: the K layout
//     vk_pack_stem_weight already produces): the B fragment of kernel row kh for 16 consecutive stem pixels is 16 B per lane
//     straight out of the staged image (pixel stride 16 B, kernel-row stride one image row), the 64 x 224 weights live in
//     REGISTERS (28 fragments) for the lifetime of the persistent workgroup; same MFMA, same K order, same epilogue arithmetic
//     as the im2col kernel -> the same bits (its eighth, all-zero K step adds +0);
//   * stem pixels go to LDS as f16 ([pixel][64 ch], 16-B chunks XOR-swizzled by pixel), pixels outside the map as -inf;
//     then every thread takes the max of 9 chunks (v_pk_max_f16) and writes 16 B of the pooled map;
//   * 256 threads, 51 KiB of LDS, <= 168 registers: three workgroups per CU overlap each other's phases.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StemPoolK {
    const char *x;           // [N, Hp, Wp, 4] f16, zero border of 3
    const char *w;           // [64][256] f16, K index = kernel row * 32 + kernel col * 4 + channel
    const float *bias;       // [64]
    char *y;                 // [N, H2, W2, 64] f16
    int Hp, Wp, H1, W1, H2, W2;
    int pad;                 // of the pool: 0 (ceil mode, windows clipped) or 1
    int tiles_x, tiles_y, ntiles;
};

constexpr int SP_PH = 4, SP_PW = 16;                    // pooled tile
constexpr int SP_SH = 2 * SP_PH + 1, SP_SW = 2 * SP_PW + 1;    // stem pixels under it: 9 x 33
constexpr int SP_NPX = SP_SH * SP_SW;                   // 297
constexpr int SP_NF = (SP_NPX + 15) / 16;               // 19 fragments of 16 pixels
constexpr int SP_IH = 2 * SP_SH + 5, SP_IW = 2 * SP_SW + 6;    // staged image: 23 x 72 pixels (kernel row reads run 8 pixels wide)
constexpr int SP_IN_BYTES = SP_IH * SP_IW * 8;          // 13 248
constexpr int SP_ST_BYTES = SP_NF * 16 * 128;           // 38 912
constexpr int SP_SMEM = SP_IN_BYTES + SP_ST_BYTES + 256;   // + the 64 bias values

__global__ __launch_bounds__(256, 3) void stem_pool_kernel(StemPoolK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *in_lds = smem, *st_lds = smem + SP_IN_BYTES;
    float *bias_lds = reinterpret_cast<float *>(smem + SP_IN_BYTES + SP_ST_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;

    // ---- weights and bias: row j of MFMA row tile ni is channel (ni>>1)*32 + (j>>2)*8 + (ni&1)*4 + (j&3), so that a lane ends
    // up with 8 consecutive channels per pair of tiles (conv_mfma.hip) ----
    half8 wf[4][7];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int row = (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) wf[ni][kh] = *reinterpret_cast<const half8 *>(p.w + row * 512 + kh * 64 + g * 16);
    }
    if (tid < 64) bias_lds[tid] = p.bias[tid];       // read per fragment from LDS (16 values per lane: no registers beside the weights')
    const half8 ninf = {(_Float16)-INFINITY, (_Float16)-INFINITY, (_Float16)-INFINITY, (_Float16)-INFINITY,
                        (_Float16)-INFINITY, (_Float16)-INFINITY, (_Float16)-INFINITY, (_Float16)-INFINITY};

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
        const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
        const int ph0 = ty * SP_PH, pw0 = tx * SP_PW;
        const int sr0 = 2 * ph0 - p.pad, sc0 = 2 * pw0 - p.pad;       // first stem pixel of the tile (may be -1)
        const int ir0 = 2 * sr0, ic0 = 2 * sc0;                       // its window's corner in the bordered image (may be -2)

        int to = tid;                                    // opaque: the per-thread offsets of the copy and pool phases are recomputed
        asm volatile("" : "+v"(to));                     // per tile instead of living in registers beside the weights
        // ---- the image under the tile -> LDS, 16 B (2 pixels) per request; outside the bordered image: zeros.  All of a thread's
        // loads are issued before the first LDS write (one memory latency per tile, not four) ----
        constexpr int NCH = SP_IH * (SP_IW / 2), NIT = (NCH + 255) / 256;
        u32x4 cv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = to + 256 * it;
            const int r = c / (SP_IW / 2), k = c - r * (SP_IW / 2);
            const int row = ir0 + r, col = ic0 + 2 * k;
            cv[it] = u32x4{0u, 0u, 0u, 0u};
            if (c < NCH && (unsigned)row < (unsigned)p.Hp && (unsigned)col < (unsigned)p.Wp)
                cv[it] = *reinterpret_cast<const u32x4 *>(p.x + (((long)n * p.Hp + row) * p.Wp + col) * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = to + 256 * it;
            const int r = c / (SP_IW / 2), k = c - r * (SP_IW / 2);
            if (c < NCH) *reinterpret_cast<u32x4 *>(in_lds + (r * SP_IW + 2 * k) * 8) = cv[it];
        }
        __syncthreads();

        // ---- conv + BN + ReLU of 16 stem pixels at a time -> f16 in LDS ----
        for (int f = wave; f < SP_NF; f += 4) {
            const int idx = f * 16 + j;
            const int idc = min(idx, SP_NPX - 1);
            const int r = idc / SP_SW, c = idc - r * SP_SW;
            const char *xa = in_lds + (2 * r * SP_IW + 2 * c) * 8 + g * 16;
            floatx4 acc[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const half8 xf = *reinterpret_cast<const half8 *>(xa + kh * (SP_IW * 8));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][kh], xf, acc[ni], 0, 0, 0);
            }
            const bool valid = idx < SP_NPX && (unsigned)(sr0 + r) < (unsigned)p.H1 && (unsigned)(sc0 + c) < (unsigned)p.W1;
            int gb = g * 8;                              // opaque: the 16 bias values are re-read (LDS) per fragment instead of held in
            asm volatile("" : "+v"(gb));                 // registers beside the 112 of the weights
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const floatx4 bq0 = *reinterpret_cast<const floatx4 *>(bias_lds + q * 32 + gb), bq1 = *reinterpret_cast<const floatx4 *>(bias_lds + q * 32 + gb + 4);
                // packed adds, v_cvt_pk_f16_f32, ReLU on the rounded halves (the same bits as max before rounding; NaN -> 0 either way)
                const floatx4 x0 = acc[2 * q] + bq0, x1 = acc[2 * q + 1] + bq1;
                const half4 h0 = __builtin_convertvector(x0, half4), h1 = __builtin_convertvector(x1, half4);
                half8 o = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                if (!valid) o = ninf;
                *reinterpret_cast<half8 *>(st_lds + idx * 128 + (((q * 4 + g) ^ (idx & 7)) << 4)) = o;
            }
        }
        __syncthreads();

        // ---- 3x3 stride-2 max over the stem pixels in LDS -> the pooled map ----
#pragma unroll
        for (int it = 0; it < SP_PH * SP_PW * 8 / 256; ++it) {
            const int item = to + 256 * it;
            const int pp = item >> 3, ch = item & 7;
            const int pr = pp / SP_PW, pc = pp - pr * SP_PW;
            half8 m = ninf;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    const int idx = (2 * pr + dr) * SP_SW + 2 * pc + dc;
                    m = __builtin_elementwise_max(m, *reinterpret_cast<const half8 *>(st_lds + idx * 128 + ((ch ^ (idx & 7)) << 4)));
                }
            const int ph = ph0 + pr, pw = pw0 + pc;
            if (ph < p.H2 && pw < p.W2) *reinterpret_cast<half8 *>(p.y + ((((long)n * p.H2 + ph) * p.W2 + pw) * 64 + ch * 8) * 2) = m;
        }
        // (no barrier here: the next tile's image goes to in_lds, which nobody reads after the second barrier above, and its
        //  stem pixels are written only after the next first barrier, which a thread reaches when its pooling reads are done)
    }
}

bool stem_pool_eligible(int cout, vk_dtype dt) {
    const char *v = getenv("VK_STEM_FUSED");             // "0": conv and pool as two kernels (A/B switch and bit-identity tests)
    if (v && v[0] == '0') return false;
    return dt == VK_F16 && cout == 64;
}

// x: stem_pack's bordered image [N, Hp, Wp, 4]; y: [N, H2, W2, 64]
int launch_stem_pool(const void *x, int N, int Hp, int Wp, int H1, int W1, const void *w, const float *bias, int caffe, void *y,
                     hipStream_t stream) {
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM));
        int dev = 0;
        hipDeviceProp_t prop;
        VK_CHECK_HIP(hipGetDevice(&dev));
        VK_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_set = true;
    }
    StemPoolK k;
    k.x = (const char *)x;
    k.w = (const char *)w;
    k.bias = bias;
    k.y = (char *)y;
    k.Hp = Hp;
    k.Wp = Wp;
    k.H1 = H1;
    k.W1 = W1;
    k.pad = caffe ? 0 : 1;
    if (caffe) {                                         // as launch_maxpool (pool.hip)
        k.H2 = (H1 - 3 + 1) / 2 + 1;
        k.W2 = (W1 - 3 + 1) / 2 + 1;
        if ((k.H2 - 1) * 2 >= H1) --k.H2;
        if ((k.W2 - 1) * 2 >= W1) --k.W2;
    } else {
        k.H2 = (H1 + 2 - 3) / 2 + 1;
        k.W2 = (W1 + 2 - 3) / 2 + 1;
    }
    VK_REQUIRE(Wp % 2 == 0 && Hp >= 2 * H1 + 5 && Wp >= 2 * W1 + 5, VK_EINVAL, "stem_pool: bordered image %dx%d too small for a %dx%d map", Hp, Wp, H1, W1);
    k.tiles_x = ceil_div(k.W2, SP_PW);
    k.tiles_y = ceil_div(k.H2, SP_PH);
    const long nt = (long)N * k.tiles_x * k.tiles_y;
    VK_REQUIRE(nt > 0 && nt < (1L << 31), VK_EINVAL, "stem_pool: %ld tiles", nt);
    k.ntiles = (int)nt;
    const int grid = (int)std::min<long>(nt, (long)n_cu * 3);
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(256), SP_SMEM, stream, k);
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        const long M = (long)N * H1 * W1;
        tm->recs.push_back({3, 2.0 * (double)M * 64 * 147, e0, e1, (int)std::min<long>(M, 1L << 30), 64, 4, 7, 2,
                            (double)N * Hp * Wp * 8 + (double)N * k.H2 * k.W2 * 128 + 64.0 * 512});
    }
    return VK_OK;
}

}  // namespace vk
