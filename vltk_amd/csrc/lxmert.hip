// Non-GEMM kernels of the LXMERT-style cross-modality encoder (the reference feeds the
// extractor's [B,36,2048] features + [B,36,4] boxes to transformers' LxmertModel, vltk/legacy/legacy_train.py:30-39).
// Restated from transformers/models/lxmert/modeling_lxmert.py (v5.15): LxmertEmbeddings :179-214, LxmertAttention
// :217-266, LayerNorm uses eps 1e-12, LxmertVisualFeatureEncoder :452-476.  The linear layers run on the MFMA GEMM of
// conv_mfma.hip (bf16 / f16 / f32); everything here is HBM- or latency-bound and computes in fp32.
//
//   layernorm_kernel   y = scale * LayerNorm(x) (+ y)      one wave per row
//   embed_ln_kernel    LayerNorm(word[id] + position[l] + token_type[tt])
//   attention_kernel   softmax(Q K^T / sqrt(d) + mask) V per (batch, head); K and V of one head live in LDS (fp32: strict mode)
//   attention_mfma_kernel  the same on v_mfma_f32_16x16x32 (bf16 / f16), one wave per (batch, head)
#include "vk_common.h"

namespace vk {

template <typename T>
__device__ __forceinline__ float ldf(const T *p) { return (float)*p; }
template <typename T>
__device__ __forceinline__ void stf(T *p, float v) { *p = (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int LN_MAX_PER_LANE = 32;      // rows up to 2048 elements

// x [M, C] (row stride ldx), y [M, C] (row stride ldy).  Mean / biased variance in fp32 over the stored values,
// (x - mean) / sqrt(var + eps) * gamma + beta like at::native::layer_norm.
template <typename T, bool EMBED>
__global__ __launch_bounds__(256) void layernorm_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, T *__restrict__ y, int ldy, int M, int C,
                                                       float eps, float scale, int accumulate,
                                                       // EMBED: x = word table, rows picked by ids; + position + token type
                                                       const int64_t *__restrict__ ids, const int64_t *__restrict__ tts,
                                                       const T *__restrict__ pos, const T *__restrict__ typ, int L) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float v[LN_MAX_PER_LANE];
    const int n = (C + 63) / 64;
    float s = 0.f;
    const T *xr = EMBED ? x + (long)ids[row] * ldx : x + (long)row * ldx;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (i < n && c < C) {
            t = ldf(xr + c);
            if (EMBED) t = t + ldf(pos + (long)(row % L) * C + c) + ldf(typ + (long)tts[row] * C + c);
        }
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (i < n && c < C) {
            const float d = v[i] - mean;
            q += d * d;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (i < n && c < C) {
            float o = ((v[i] - mean) * rstd * gamma[c] + beta[c]) * scale;
            T *yp = y + (long)row * ldy + c;
            if (accumulate) o += ldf(yp);
            stf(yp, o);
        }
    }
}

// 16-byte-vector form of layernorm_kernel for rows whose length is a multiple of 8 elements (2-byte types) / 4 (f32): a lane owns
// chunks lane, lane + 64, ... of the row.  Same arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, T *__restrict__ y, int ldy, int M, int C,
                                                           float eps, float scale, int accumulate) {
    constexpr int V = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    constexpr int MAXC = LN_MAX_PER_LANE / V;          // chunks per lane
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int chunks = C / V;
    float v[MAXC][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < chunks) {
            const vecT t = *reinterpret_cast<const vecT *>(x + (long)row * ldx + ch * V);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                v[i][e] = (float)t[e];
                s += v[i][e];
            }
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
        if (lane + 64 * i < chunks) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < chunks) {
            T *yp = y + (long)row * ldy + ch * V;
            vecT o;
            vecT old;
            if (accumulate) old = *reinterpret_cast<const vecT *>(yp);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float r = ((v[i][e] - mean) * rstd * gamma[ch * V + e] + beta[ch * V + e]) * scale;
                if (accumulate) r += (float)old[e];
                o[e] = (T)r;
            }
            *reinterpret_cast<vecT *>(yp) = o;
        }
    }
}

// One workgroup per (batch b, head h).  Q, K, V of the head are staged in LDS as fp32 (rows padded by one float so
// that column walks do not collide on a bank); 256 threads share the Lq x Lk scores, one wave per row does the
// soft-max, then the threads share the Lq x d outputs.  Scores, soft-max and the weighted sum are fp32 (the
// reference matmuls and soft-maxes in the model dtype; the bf16-emulating oracle restates THIS kernel's rounding
// points: inputs and output only).
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ k, int ldk,
                                                       const T *__restrict__ v, int ldv, const float *__restrict__ mask,
                                                       T *__restrict__ out, int ldo, int heads, int Lq, int Lk, int d, float scale) {
    extern __shared__ float sm[];
    const int dp = d + 1, lp = Lk + 1;
    float *qs = sm, *ks = qs + Lq * dp, *vs = ks + Lk * dp, *ps = vs + Lk * d;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    constexpr int V = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    if (d % V == 0 && ldq % V == 0 && ldk % V == 0 && ldv % V == 0) {       // 16-byte staging loads
        const int dv = d / V;
        for (int i = tid; i < Lk * dv; i += 256) {
            const int r = i / dv, c = (i - r * dv) * V;
            const vecT kk = *reinterpret_cast<const vecT *>(k + ((long)b * Lk + r) * ldk + h * d + c);
            const vecT vv = *reinterpret_cast<const vecT *>(v + ((long)b * Lk + r) * ldv + h * d + c);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                ks[r * dp + c + e] = (float)kk[e];
                vs[r * d + c + e] = (float)vv[e];
            }
        }
        for (int i = tid; i < Lq * dv; i += 256) {
            const int r = i / dv, c = (i - r * dv) * V;
            const vecT qq = *reinterpret_cast<const vecT *>(q + ((long)b * Lq + r) * ldq + h * d + c);
#pragma unroll
            for (int e = 0; e < V; ++e) qs[r * dp + c + e] = (float)qq[e];
        }
    } else {
        for (int i = tid; i < Lk * d; i += 256) {
            const int r = i / d, c = i - r * d;
            ks[r * dp + c] = ldf(k + ((long)b * Lk + r) * ldk + h * d + c);
            vs[i] = ldf(v + ((long)b * Lk + r) * ldv + h * d + c);
        }
        for (int i = tid; i < Lq * d; i += 256) {
            const int r = i / d, c = i - r * d;
            qs[r * dp + c] = ldf(q + ((long)b * Lq + r) * ldq + h * d + c);
        }
    }
    __syncthreads();
    for (int i = tid; i < Lq * Lk; i += 256) {
        const int r = i / Lk, j = i - r * Lk;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += qs[r * dp + c] * ks[j * dp + c];
        ps[r * lp + j] = s * scale + (mask ? mask[(long)b * Lk + j] : 0.f);
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < Lq; r += 4) {
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, ps[r * lp + j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float den = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float e = expf(ps[r * lp + j] - mx);
            ps[r * lp + j] = e;
            den += e;
        }
        den = wave_sum(den);
        const float inv = 1.0f / den;
        for (int j = lane; j < Lk; j += 64) ps[r * lp + j] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < Lq * d; i += 256) {
        const int r = i / d, c = i - r * d;
        float o = 0.f;
        for (int j = 0; j < Lk; ++j) o += ps[r * lp + j] * vs[j * d + c];
        stf(out + ((long)b * Lq + r) * ldo + h * d + c, o);
    }
}

// ---- the same attention on the matrix cores (bf16 / f16, head dim 32 or 64, Lq <= 48, Lk <= 64: LXMERT runs 20 text
// and 36 visual tokens) ----
// One WAVE per (batch, head), four per workgroup.  S = Q K^T and O = P V are v_mfma_f32_16x16x32 tiles:
//   * Q and K rows are the A / B operand fragments as they lie in memory (8 consecutive head-dim elements per lane):
//     no staging at all for the first product;
//   * a score tile leaves a lane with S[q = 4g + e][key = j]: the soft-max over the keys of a row reduces over the
//     key blocks in registers and over the 16 lanes of a group with four shuffles; fp32 throughout;
//   * P (rounded to the storage type, as the reference's own bf16 matmul sees it) and V^T go through a per-wave LDS
//     image to become the second product's operands; padded keys carry P = 0 and V = 0.
typedef _Float16 att_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_b8 __attribute__((ext_vector_type(8)));
typedef float att_f4 __attribute__((ext_vector_type(4)));
template <typename T>
struct AttFrag;
template <>
struct AttFrag<_Float16> {
    typedef att_h8 type;
    static __device__ __forceinline__ att_f4 mma(att_h8 a, att_h8 b, att_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct AttFrag<__bf16> {
    typedef att_b8 type;
    static __device__ __forceinline__ att_f4 mma(att_b8 a, att_b8 b, att_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr int ATT_MAXQB = 3, ATT_MAXKB = 4;     // 48 queries, 64 keys
constexpr int ATT_PP = 72;                      // row pitch (elements) of the P and V^T images: 64 keys + 8 (bank skew)

template <typename T, int D>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ k, int ldk,
                                                            const T *__restrict__ v, int ldv, const float *__restrict__ mask,
                                                            T *__restrict__ out, int ldo, int heads, int Lq, int Lk, int total,
                                                            float scale) {
    typedef typename AttFrag<T>::type frag;
    constexpr int KD = D / 32, NDB = D / 16;
    __shared__ __attribute__((aligned(16))) T lds[4][(48 + D) * ATT_PP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int pair = blockIdx.x * 4 + wave;
    const bool live = pair < total;
    const int pc = live ? pair : total - 1;             // idle waves redo the last pair (no early exit before the barrier)
    const int b = pc / heads, h = pc - b * heads;
    const int QB = (Lq + 15) >> 4, KB = (Lk + 15) >> 4, KS2 = (Lk + 31) >> 5;
    T *pt = lds[wave], *vt = pt + 48 * ATT_PP;
    const frag zf = {};

    // ---- operand fragments straight from memory ----
    frag qf[ATT_MAXQB][KD], kf[ATT_MAXKB][KD];
#pragma unroll
    for (int qb = 0; qb < ATT_MAXQB; ++qb) {
        const int r = qb * 16 + j;
        const bool ok = qb < QB && r < Lq;
        const T *src = q + ((long)b * Lq + (ok ? r : 0)) * ldq + h * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const frag t = *reinterpret_cast<const frag *>(src + ks * 32);
            qf[qb][ks] = ok ? t : zf;
        }
    }
#pragma unroll
    for (int kb = 0; kb < ATT_MAXKB; ++kb) {
        const int r = kb * 16 + j;
        const bool ok = kb < KB && r < Lk;
        const T *src = k + ((long)b * Lk + (ok ? r : 0)) * ldk + h * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            const frag t = *reinterpret_cast<const frag *>(src + ks * 32);
            kf[kb][ks] = ok ? t : zf;
        }
    }
    // ---- V^T image: vt[d][key], zero for the padded keys ----
    {
        constexpr int CH = D / 8;                        // 16-byte chunks per V row
        for (int i = lane; i < KS2 * 32 * CH; i += 64) {
            const int key = i / CH, c = i - key * CH;
            frag t = zf;
            if (key < Lk) t = *reinterpret_cast<const frag *>(v + ((long)b * Lk + key) * ldv + h * D + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(c * 8 + e) * ATT_PP + key] = t[e];
        }
    }
    // ---- S = Q K^T * scale + mask ----
    att_f4 sc[ATT_MAXQB][ATT_MAXKB];
#pragma unroll
    for (int qb = 0; qb < ATT_MAXQB; ++qb)
#pragma unroll
        for (int kb = 0; kb < ATT_MAXKB; ++kb) {
            att_f4 a = {0.f, 0.f, 0.f, 0.f};
            if (qb < QB && kb < KB) {
#pragma unroll
                for (int ks = 0; ks < KD; ++ks) a = AttFrag<T>::mma(qf[qb][ks], kf[kb][ks], a);
            }
            const int key = kb * 16 + j;
            const float m = key < Lk ? (mask ? mask[(long)b * Lk + key] : 0.f) : -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = key < Lk ? a[e] * scale + m : -INFINITY;
            sc[qb][kb] = a;
        }
    // ---- soft-max over the keys of each row q = qb*16 + 4g + e: registers (kb), then the 16 lanes of the group ----
#pragma unroll
    for (int qb = 0; qb < ATT_MAXQB; ++qb) {
        if (qb >= QB) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mx = sc[qb][0][e];
#pragma unroll
            for (int kb = 1; kb < ATT_MAXKB; ++kb) mx = fmaxf(mx, sc[qb][kb][e]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float den = 0.f;
#pragma unroll
            for (int kb = 0; kb < ATT_MAXKB; ++kb) {
                const float ex = expf(sc[qb][kb][e] - mx);       // padded keys: exp(-inf) = 0
                sc[qb][kb][e] = ex;
                den += ex;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
            const float inv = 1.0f / den;
            const int row = qb * 16 + g * 4 + e;
#pragma unroll
            for (int kb = 0; kb < ATT_MAXKB; ++kb) pt[row * ATT_PP + kb * 16 + j] = (T)(sc[qb][kb][e] * inv);
        }
    }
    __syncthreads();
    // ---- O = P V: A = P rows (keys contiguous), B = V^T rows (keys contiguous) ----
#pragma unroll
    for (int qb = 0; qb < ATT_MAXQB; ++qb) {
        if (qb >= QB) continue;
        frag pf[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) pf[k2] = *reinterpret_cast<const frag *>(pt + (qb * 16 + j) * ATT_PP + k2 * 32 + g * 8);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            att_f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                if (k2 >= KS2) continue;
                const frag vf = *reinterpret_cast<const frag *>(vt + (db * 16 + j) * ATT_PP + k2 * 32 + g * 8);
                o = AttFrag<T>::mma(pf[k2], vf, o);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = qb * 16 + g * 4 + e;
                if (live && row < Lq) out[((long)b * Lq + row) * ldo + h * D + db * 16 + j] = (T)o[e];
            }
        }
    }
}

template <typename T>
static int ln_launch(const void *x, int ldx, const float *g, const float *b, void *y, int ldy, int M, int C, float eps, float scale,
                     int accumulate, const int64_t *ids, const int64_t *tts, const void *pos, const void *typ, int L, hipStream_t s) {
    const dim3 grid((M + 3) / 4), block(256);
    constexpr int V = 16 / sizeof(T);
    if (!ids && C % V == 0 && ldx % V == 0 && ldy % V == 0) {
        hipLaunchKernelGGL((layernorm_vec_kernel<T>), grid, block, 0, s, (const T *)x, ldx, g, b, (T *)y, ldy, M, C, eps, scale, accumulate);
        VK_CHECK_HIP(hipGetLastError());
        return VK_OK;
    }
    if (ids)
        hipLaunchKernelGGL((layernorm_kernel<T, true>), grid, block, 0, s, (const T *)x, ldx, g, b, (T *)y, ldy, M, C, eps, scale, accumulate,
                           ids, tts, (const T *)pos, (const T *)typ, L);
    else
        hipLaunchKernelGGL((layernorm_kernel<T, false>), grid, block, 0, s, (const T *)x, ldx, g, b, (T *)y, ldy, M, C, eps, scale, accumulate,
                           nullptr, nullptr, nullptr, nullptr, 1);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

static int ln_dispatch(vk_dtype dt, const void *x, int ldx, const float *g, const float *b, void *y, int ldy, int M, int C, float eps,
                       float scale, int accumulate, const int64_t *ids, const int64_t *tts, const void *pos, const void *typ, int L,
                       hipStream_t s) {
    VK_REQUIRE(M > 0 && C > 0 && C <= 64 * LN_MAX_PER_LANE, VK_EINVAL, "layernorm: C=%d must be in 1..%d", C, 64 * LN_MAX_PER_LANE);
    switch (dt) {
        case VK_F32: return ln_launch<float>(x, ldx, g, b, y, ldy, M, C, eps, scale, accumulate, ids, tts, pos, typ, L, s);
        case VK_F16: return ln_launch<_Float16>(x, ldx, g, b, y, ldy, M, C, eps, scale, accumulate, ids, tts, pos, typ, L, s);
        case VK_BF16: return ln_launch<__bf16>(x, ldx, g, b, y, ldy, M, C, eps, scale, accumulate, ids, tts, pos, typ, L, s);
        default: break;
    }
    VK_REQUIRE(false, VK_EINVAL, "layernorm: dtype must be f32, f16 or bf16");
    return VK_EINVAL;
}

}  // namespace vk

using namespace vk;

extern "C" {

int vk_layernorm(const void *x, int ldx, const float *gamma, const float *beta, void *y, int ldy, int M, int C, float eps, float scale,
                 int accumulate, vk_dtype dt, void *stream) {
    VK_REQUIRE(x && gamma && beta && y, VK_EINVAL, "layernorm: null argument");
    return ln_dispatch(dt, x, ldx, gamma, beta, y, ldy, M, C, eps, scale, accumulate, nullptr, nullptr, nullptr, nullptr, 1,
                       (hipStream_t)stream);
}

int vk_embed_layernorm(const int64_t *input_ids, const int64_t *token_type_ids, int B, int L, const void *word, const void *position,
                       const void *token_type, const float *gamma, const float *beta, void *y, int C, float eps, vk_dtype dt,
                       void *stream) {
    VK_REQUIRE(input_ids && token_type_ids && word && position && token_type && gamma && beta && y, VK_EINVAL, "embed: null argument");
    return ln_dispatch(dt, word, C, gamma, beta, y, C, B * L, C, eps, 1.0f, 0, input_ids, token_type_ids, position, token_type, L,
                       (hipStream_t)stream);
}

int vk_attention(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const float *mask, void *out, int ldo, int B,
                 int heads, int Lq, int Lk, int d, vk_dtype dt, void *stream) {
    VK_REQUIRE(q && k && v && out, VK_EINVAL, "attention: null argument");
    VK_REQUIRE(B > 0 && heads > 0 && Lq >= 1 && Lk >= 1 && d >= 1, VK_EINVAL, "attention: bad geometry (Lq=%d Lk=%d d=%d)", Lq, Lk, d);
    const size_t smem = ((size_t)(Lq + Lk) * (d + 1) + (size_t)Lk * d + (size_t)Lq * (Lk + 1)) * sizeof(float);
    VK_REQUIRE(smem <= 160 * 1024, VK_EINVAL,
               "attention: one head's Q, K, V and scores (%zu bytes at Lq=%d Lk=%d d=%d) must fit the 160 KiB LDS", smem, Lq, Lk, d);
    const float scale = 1.0f / sqrtf((float)d);
    hipStream_t s = (hipStream_t)stream;
    // matrix-core form: 16-bit types, head dim 32 / 64, up to 48 queries and 64 keys, 16-byte aligned rows
    static const bool no_mfma = getenv("VK_ATTENTION_MFMA") && getenv("VK_ATTENTION_MFMA")[0] == '0';     // A/B switch
    if (!no_mfma && (dt == VK_F16 || dt == VK_BF16) && (d == 32 || d == 64) && Lq <= 48 && Lk <= 64 && ldq % 8 == 0 && ldk % 8 == 0 &&
        ldv % 8 == 0 && ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0) {
        const int total = B * heads;
        const dim3 grid((total + 3) / 4), block(256);
#define VK_ATTM(T, DD)                                                                                                   \
    hipLaunchKernelGGL((attention_mfma_kernel<T, DD>), grid, block, 0, s, (const T *)q, ldq, (const T *)k, ldk, (const T *)v, ldv, mask, \
                       (T *)out, ldo, heads, Lq, Lk, total, scale)
        if (dt == VK_F16) {
            if (d == 64) VK_ATTM(_Float16, 64); else VK_ATTM(_Float16, 32);
        } else {
            if (d == 64) VK_ATTM(__bf16, 64); else VK_ATTM(__bf16, 32);
        }
#undef VK_ATTM
        VK_CHECK_HIP(hipGetLastError());
        return VK_OK;
    }
    const dim3 grid(B * heads), block(256);
#define VK_ATT(T)                                                                                                        \
    do {                                                                                                                 \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<T>),                        \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL(attention_kernel<T>, grid, block, smem, s, (const T *)q, ldq, (const T *)k, ldk, (const T *)v, ldv, mask, \
                           (T *)out, ldo, heads, Lq, Lk, d, scale);                                                     \
    } while (0)
    switch (dt) {
        case VK_F32: VK_ATT(float); break;
        case VK_F16: VK_ATT(_Float16); break;
        case VK_BF16: VK_ATT(__bf16); break;
        default: VK_REQUIRE(false, VK_EINVAL, "attention: dtype must be f32, f16 or bf16");
    }
#undef VK_ATT
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // extern "C"
