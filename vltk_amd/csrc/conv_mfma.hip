// Convolution (+ folded BatchNorm bias, + residual, + ReLU) as an implicit GEMM on the
// CDNA4 matrix cores.  Replaces Conv2d.forward (reference vltk/modeling/frcnn.py:794-822)
// and the conv/BN/ReLU/`+= shortcut` sequence of BottleneckBlock.forward (:963-979),
// BasicStem.forward (:872-874) and RPNHead.forward (:1569-1571); also used as the plain
// GEMM of FastRCNNOutputLayers (:1729-1737).
//
// Layout (HBM): activations NHWC, i.e. a row-major [M = N*Ho*Wo, C] matrix; weights packed
// [Cout_pad][K] with K = (kh, kw, cin) contiguous per output channel and padded to whole
// 128-byte K-tiles.  GEMM view: Y[m][co] = sum_k X_im2col[m][k] * Wt[co][k].
//
// Tiling: one 256-thread workgroup (4 waves, 2x2) computes a 128 (pixels) x BN (channels)
// tile; K advances in 128-byte tiles (64 f16 / 32 f32 elements) through a double-buffered,
// XOR-swizzled LDS image (row = 128 B = 8 x 16-B chunks, chunk' = chunk ^ (row & 7):
// conflict-free ds_read_b128 for the 16x16x32 operand lane map).  Global->LDS staging goes
// through registers so that out-of-image taps are zero-filled and the next K-tile's loads
// are in flight while the current one is multiplied (one barrier per K-tile).
// The weight fragment is the MFMA "A" operand and the pixel fragment the "B" operand, so each
// lane ends up with consecutive output CHANNELS of one pixel; two 16-wide channel tiles are
// interleaved in the fragment addressing so a lane owns 8 consecutive channels = one 16-byte
// store (f16) of the NHWC output row.
//
// fp16 mode: v_mfma_f32_16x16x32_f16, fp32 accumulate.  fp32 ("strict") mode:
// v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) on the same loader/epilogue.
#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct ConvK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int H, W, Ho, Wo, HoWo, M;
    int cin_bytes;       // Cin * sizeof(T)
    int cout8;           // output channels rounded up to 8 (store bound)
    int ldy;
    int kw, stride, pad, dil;
    int ktiles, kt_per_tap;
    int concurrent;      // timing bucket 6 (two-stream section)
    int slice_bytes;     // grouped conv: bytes of the input-channel slice a 64-channel output tile reads (0: dense)
    int wrow_bytes;      // ktiles * 128
    int relu;
    int m_tiles, n_tiles;
    double alg_flops;    // 2 * M * Cout * (kh*kw*Cin): host-side bookkeeping only
    double alg_bytes;    // input + output (+ residual) + weights, each once
};

template <typename T>
struct Frag;
template <>
struct Frag<_Float16> {
    typedef half8 type;
    static constexpr int KSTEPS = 2;  // 64 halfs per K-tile, 32 per MFMA
};
template <>
struct Frag<__bf16> {
    typedef bf16x8 type;
    static constexpr int KSTEPS = 2;  // 64 bf16 per K-tile, 32 per MFMA
};
template <>
struct Frag<float> {
    typedef float type;
    static constexpr int KSTEPS = 8;  // 32 floats per K-tile, 4 per MFMA
};

__device__ __forceinline__ floatx4 mfma(half8 a, half8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 mfma(bf16x8 a, bf16x8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 mfma(float a, float b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename Frag<T>::type lds_frag(const char *tile, int row, int ks, int g);
template <>
__device__ __forceinline__ half8 lds_frag<_Float16>(const char *tile, int row, int ks, int g) {
    int chunk = ks * 4 + g;
    return *reinterpret_cast<const half8 *>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
template <>
__device__ __forceinline__ bf16x8 lds_frag<__bf16>(const char *tile, int row, int ks, int g) {
    int chunk = ks * 4 + g;
    return *reinterpret_cast<const bf16x8 *>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
template <>
__device__ __forceinline__ float lds_frag<float>(const char *tile, int row, int ks, int g) {
    return *reinterpret_cast<const float *>(tile + row * 128 + ((ks ^ (row & 7)) << 4) + g * 4);
}

template <typename OutT>
__device__ __forceinline__ void store8(char *dst, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<_Float16>(char *dst, const float (&v)[8]) {
    half8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
    *reinterpret_cast<half8 *>(dst) = h;
}
template <>
__device__ __forceinline__ void store8<__bf16>(char *dst, const float (&v)[8]) {
    bf16x8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)v[i];     // round to nearest even
    *reinterpret_cast<bf16x8 *>(dst) = h;
}
template <>
__device__ __forceinline__ void store8<float>(char *dst, const float (&v)[8]) {
    floatx4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    reinterpret_cast<floatx4 *>(dst)[0] = a;
    reinterpret_cast<floatx4 *>(dst)[1] = b;
}

template <typename T>
__device__ __forceinline__ void load8(const char *src, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<_Float16>(const char *src, float (&v)[8]) {
    half8 h = *reinterpret_cast<const half8 *>(src);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <>
__device__ __forceinline__ void load8<__bf16>(const char *src, float (&v)[8]) {
    bf16x8 h = *reinterpret_cast<const bf16x8 *>(src);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <>
__device__ __forceinline__ void load8<float>(const char *src, float (&v)[8]) {
    floatx4 a = reinterpret_cast<const floatx4 *>(src)[0], b = reinterpret_cast<const floatx4 *>(src)[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
}

template <typename T, typename OutT, int BN, bool STEM>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = sizeof(T);
    constexpr int A_BYTES = CONV_BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NI = BN / 32;        // 16-wide channel tiles per wave
    constexpr int B_ITERS = BN / 32;   // 16-B chunks of the weight tile per thread
    constexpr int KSTEPS = Frag<T>::KSTEPS;
    typedef typename Frag<T>::type frag_t;

    // XCD-aware (bijective) workgroup -> tile map: the 8 XCDs each get a contiguous range of
    // tiles, and within it tiles that share the same 128 pixels (different channel tiles) are
    // adjacent, so the pixel panel is fetched from HBM once per XCD L2.
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tile = t % p.n_tiles, m_tile = t / p.n_tiles;
    const int m0 = m_tile * CONV_BM, n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int chunk = tid & 7;       // 16-B chunk of the 128-B K-tile row this thread stages
    const int lrow = tid >> 3;       // + 32*i

    // ---- per-thread im2col row state (4 pixel rows) ----
    long a_off[4];
    int bh[4], bw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + lrow + 32 * i;
        bool ok = m < p.M;
        int mm = ok ? m : 0;
        int n_img = mm / p.HoWo;
        int rem = mm - n_img * p.HoWo;
        int ho = rem / p.Wo;
        int wo = rem - ho * p.Wo;
        int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        a_off[i] = ((long)(n_img * p.H + h0) * p.W + w0) * p.cin_bytes;
        bh[i] = ok ? h0 : -(1 << 28);   // invalid rows fail every bounds test below
        bw[i] = w0;
    }
    const char *wsrc[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i)
        wsrc[i] = p.w + (long)(n0 + lrow + 32 * i) * p.wrow_bytes + chunk * 16;

    u32x4 ra[4], rb[B_ITERS];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // (khi, kwi, cb) describe K-tile `kt`: the tap and the byte offset inside the tap's channel run.
    auto load_tile = [&](int kt, int khi, int kwi, int cb) {
        if constexpr (STEM) {
            // K-tile = runs of 32 consecutive elements of one padded-image row starting at the
            // window's left edge (7 taps x 4 channels + 4 zero-weighted elements) per kernel row.
            constexpr int CPR = 2 * ES;             // chunks per run (64 B f16 / 128 B f32)
            const int kr = kt * (8 / CPR) + chunk / CPR;   // kernel row
            const int inner = (chunk % CPR) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool ok = (bh[i] >= 0) && (kr < 7);
                const char *src = p.x + a_off[i] + (long)kr * p.W * p.cin_bytes + inner;
                ra[i] = ok ? *reinterpret_cast<const u32x4 *>(src) : zero4;
            }
        } else {
            const int dh = khi * p.dil, dw = kwi * p.dil;
            // grouped: the tile's 64 output channels only see the slice of input channels their groups own
            const int sl = p.slice_bytes ? (n0 * ES / p.slice_bytes) * p.slice_bytes : 0;
            const long toff = ((long)dh * p.W + dw) * p.cin_bytes + sl + cb + chunk * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool ok = (unsigned)(bh[i] + dh) < (unsigned)p.H && (unsigned)(bw[i] + dw) < (unsigned)p.W;
                const char *src = p.x + a_off[i] + toff;
                ra[i] = ok ? *reinterpret_cast<const u32x4 *>(src) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) rb[i] = *reinterpret_cast<const u32x4 *>(wsrc[i] + (long)kt * 128);
    };
    auto store_tile = [&](int stage) {
        char *sa = smem + stage * STAGE;
        char *sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = lrow + 32 * i;
            *reinterpret_cast<u32x4 *>(sa + row * 128 + ((chunk ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            int row = lrow + 32 * i;
            *reinterpret_cast<u32x4 *>(sb + row * 128 + ((chunk ^ (row & 7)) << 4)) = rb[i];
        }
    };

    // ---- fragment addressing ----
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, j = lane & 15;
    int xrow[4], wrow[NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) xrow[mi] = wm * 64 + mi * 16 + j;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        wrow[ni] = wn * (BN / 2) + (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);

    floatx4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop ----
    int khi = 0, kwi = 0, ktt = 0;   // tap state of the NEXT tile to load
    auto advance = [&]() {
        if (++ktt == p.kt_per_tap) {
            ktt = 0;
            if (++kwi == p.kw) {
                kwi = 0;
                ++khi;
            }
        }
    };
    load_tile(0, khi, kwi, 0);
    advance();
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < p.ktiles; ++kt) {
        const bool more = kt + 1 < p.ktiles;
        if (more) {
            load_tile(kt + 1, khi, kwi, ktt * 128);
            advance();
        }
        const char *sa = smem + (kt & 1) * STAGE;
        const char *sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            frag_t xf[4], wf[NI];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xf[mi] = lds_frag<T>(sa, xrow[mi], ks, g);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni] = lds_frag<T>(sb, wrow[ni], ks, g);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma(wf[ni], xf[mi], acc[mi][ni]);
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual) (ReLU) -> OutT, 8 consecutive channels per lane ----
    // residual loads of a channel group are issued together (clamped row, predicated store)
#pragma unroll
    for (int qn = 0; qn < NI / 2; ++qn) {
        const int co = n0 + wn * (BN / 2) + qn * 32 + g * 8;
        if (co >= p.cout8) continue;
        float b[8];
        load8<float>(reinterpret_cast<const char *>(p.bias + co), b);
        float rr[4][8];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (p.res) {
                const int m = min(m0 + wm * 64 + mi * 16 + j, p.M - 1);
                load8<T>(p.res + ((long)m * p.ldy + co) * ES, rr[mi]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) rr[mi][e] = 0.f;
            }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + j;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[mi][2 * qn][e] + b[e] + rr[mi][e];
                v[4 + e] = acc[mi][2 * qn + 1][e] + b[4 + e] + rr[mi][4 + e];
            }
            if (p.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            } else if (p.relu == 2) {      // GELU, erf form (transformers ACT2FN["gelu"])
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
            } else if (p.relu == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
            }
            if (m < p.M) store8<OutT>(p.y + ((long)m * p.ldy + co) * (long)sizeof(OutT), v);
        }
    }
}

template <typename T, typename OutT, int BN, bool STEM>
static int launch_t(const ConvK &k, hipStream_t stream) {
    constexpr int smem = 2 * (CONV_BM * 128 + BN * 128);
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<T, OutT, BN, STEM>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid(k.m_tiles * k.n_tiles), block(256);
    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL((conv_mfma_kernel<T, OutT, BN, STEM>), grid, block, smem, stream, k);
    VK_CHECK_HIP(hipGetLastError());
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        int bucket = 3;
        if (sizeof(T) == 2 && !STEM) bucket = sizeof(OutT) == 4 ? 2 : 1;
        if (k.concurrent) bucket = 6;
        tm->recs.push_back({bucket, k.alg_flops, e0, e1, k.M, k.cout8, k.cin_bytes / (int)sizeof(T), k.ktiles / k.kt_per_tap, k.stride, k.alg_bytes});
    }
    return VK_OK;
}

int launch_conv(const ConvArgs &a, hipStream_t stream) {
    const bool grouped = a.groups > 1;
    if (a.pool_part) {
        VK_REQUIRE(conv_duo_pool_ok(a) && (!a.x2 || conv_duo_dual_ok(a)) && a.relu <= 1 && a.ldy == a.Cout, VK_EINVAL,
                   "conv: the fused-mean form is 1x1, stride 1, f16, Cout %% 256 == 0, Ho*Wo >= 128");
        // per-image fp64 sums (conv_ws) or per-tile integer partials (two-per-CU kernel): launch_pool_finish asks the same function
        if (conv_ws_pool_ok(a)) return launch_conv_ws(a, stream);
        return launch_conv_duo(a, stream);
    }
    if (a.x2) {
        if (conv_ws_eligible(a)) return launch_conv_ws(a, stream);           // 64 + 64 channels (res2's first conv3 + shortcut)
        if (conv_gemm4_eligible(a)) return launch_conv_gemm4(a, stream);
        if (conv256_dual_ok(a)) return launch_conv256(a, stream);
        VK_REQUIRE(conv_duo_dual_ok(a), VK_EINVAL,
                   "conv: the dual-source form is 1x1, stride 1, f16, Cout %% 256 == 0, Cin and Cin2 multiples of 32");
        return launch_conv_duo(a, stream);
    }
    if (conv3x3_blk_eligible(a)) return launch_conv3x3_blk(a, stream);
    if (!grouped) {
        if (conv3x3_panel_eligible(a)) return launch_conv3x3_panel(a, stream);
        if (conv_ws_eligible(a)) return launch_conv_ws(a, stream);
        if (conv_gemm4_eligible(a)) return launch_conv_gemm4(a, stream);
        if (conv_duo_eligible(a)) return launch_conv_duo(a, stream);
        if (conv256_eligible(a)) return launch_conv256(a, stream);
    }
    const int es = (int)dtype_size(a.dt);
    VK_REQUIRE(a.dt == VK_F16 || a.dt == VK_F32 || a.dt == VK_BF16, VK_EINVAL, "conv: dtype must be f16, bf16 or f32");
    VK_REQUIRE(a.out_dt == a.dt || a.out_dt == VK_F32, VK_EINVAL, "conv: out dtype must equal dtype or be f32");
    ConvK k;
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
    long M = (long)a.N * a.Ho * a.Wo;
    VK_REQUIRE(M > 0 && M < (1L << 31) - CONV_BM, VK_EINVAL, "conv: M=%ld out of range", M);
    k.M = (int)M;
    k.cin_bytes = a.Cin * es;
    k.cout8 = (a.Cout + 7) / 8 * 8;
    VK_REQUIRE(a.ldy >= k.cout8 && a.ldy % 8 == 0, VK_EINVAL, "conv: ldy=%d must be >= %d and a multiple of 8", a.ldy, k.cout8);
    k.ldy = a.ldy;
    k.kw = a.kw;
    k.stride = a.stride;
    k.pad = a.pad;
    k.dil = a.dil;
    k.relu = a.relu;
    k.concurrent = a.concurrent;
    k.slice_bytes = 0;
    if (a.stem) {
        VK_REQUIRE(a.Cin == 4 && a.kh == 7 && a.kw == 7 && a.stride == 2, VK_EINVAL, "conv: stem mode is 7x7 s2 on NHWC4");
        k.pad = 0;
        k.kt_per_tap = 1;
        k.ktiles = (es == 2) ? 4 : 7;
    } else if (grouped) {
        // slice-diagonal GEMM (vk_pack_conv_weight): every 64-channel output tile multiplies the input-channel
        // slice its groups own by weights that are zero outside each channel's group
        VK_REQUIRE(a.Cin == a.Cout && a.Cin % a.groups == 0, VK_EINVAL, "conv: grouped needs Cin == Cout, Cin %% groups == 0");
        const int sw = vk_conv_slice_channels(a.Cin, a.groups);
        VK_REQUIRE(sw > 0 && (sw * es) % CONV_KTILE_BYTES == 0, VK_EINVAL,
                   "conv: grouped slice of %d channels is not a whole number of K-tiles for this dtype", sw);
        k.slice_bytes = sw * es;
        k.kt_per_tap = k.slice_bytes / CONV_KTILE_BYTES;
        k.ktiles = a.kh * a.kw * k.kt_per_tap;
    } else {
        VK_REQUIRE(k.cin_bytes % CONV_KTILE_BYTES == 0, VK_EINVAL,
                   "conv: Cin=%d must be a multiple of %d for this dtype", a.Cin, CONV_KTILE_BYTES / es);
        k.kt_per_tap = k.cin_bytes / CONV_KTILE_BYTES;
        k.ktiles = a.kh * a.kw * k.kt_per_tap;
    }
    k.wrow_bytes = k.ktiles * CONV_KTILE_BYTES;
    k.alg_flops = 2.0 * (double)k.M * a.Cout * (a.stem ? 147.0 : (double)a.kh * a.kw * a.Cin / (grouped ? a.groups : 1));
    k.alg_bytes = (double)a.N * a.H * a.W * a.Cin * es + (double)k.M * a.Cout * (dtype_size(a.out_dt) + (a.res ? es : 0)) +
                  (double)a.Cout * a.kh * a.kw * a.Cin / (grouped ? a.groups : 1) * es;
    k.m_tiles = ceil_div(k.M, CONV_BM);
    const bool narrow = a.Cout <= 64 || a.stem || grouped;
    k.n_tiles = ceil_div(a.Cout, narrow ? 64 : 128);
    const bool f32out = (a.out_dt == VK_F32);
    if (a.dt == VK_BF16) {      // LXMERT-style encoder GEMMs (N3): same tiling, v_mfma_f32_16x16x32_bf16
        VK_REQUIRE(!a.stem, VK_EINVAL, "conv: no bf16 stem");
        if (f32out) return narrow ? launch_t<__bf16, float, 64, false>(k, stream) : launch_t<__bf16, float, 128, false>(k, stream);
        return narrow ? launch_t<__bf16, __bf16, 64, false>(k, stream) : launch_t<__bf16, __bf16, 128, false>(k, stream);
    }
    if (a.dt == VK_F16) {
        if (a.stem) return launch_t<_Float16, _Float16, 64, true>(k, stream);
        if (f32out) return narrow ? launch_t<_Float16, float, 64, false>(k, stream)
                                  : launch_t<_Float16, float, 128, false>(k, stream);
        return narrow ? launch_t<_Float16, _Float16, 64, false>(k, stream)
                      : launch_t<_Float16, _Float16, 128, false>(k, stream);
    }
    if (a.stem) return launch_t<float, float, 64, true>(k, stream);
    return narrow ? launch_t<float, float, 64, false>(k, stream) : launch_t<float, float, 128, false>(k, stream);
}

}  // namespace vk
