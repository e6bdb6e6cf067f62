// 1x1 convolution with K <= 512 input channels as a WEIGHT-STATIONARY GEMM: the conv3 layers of the bottlenecks
// (`out = conv3(out) + shortcut`, reference vltk/modeling/frcnn.py:970-979: 512 -> 2048 in the Res5 head, 256 -> 1024 in
// res4, 128 -> 512 in res3).
//
// Why: these layers are bound by what a CU can move through its vector-memory path (~21 B/clk measured), not by
// the matrix cores and not by HBM.  A 128 x 256 tile of the two-per-CU kernel (conv_mfma_duo.hip) moves, per output element
// at K = 512, 4 B of pixels + 8 B of WEIGHTS + 4 B of residual / output; the weights are the same 256 x K block for every
// tile of a column, re-streamed from L2 each time.  Here a workgroup OWNS one 256-channel column block for its whole life:
//   * its weights (256 x K f16 = 256 KiB at K = 512) are loaded once into REGISTERS -- each of the four waves keeps the
//     fragments of its 64 channels for every K step (64 x 4 VGPRs) -- so the K loop streams pixels only: 8 B per output
//     element instead of 16;
//   * pixels run through a 6-slot LDS ring of 64 rows x 128 channels (16 KiB) filled by LDS-DMA five stages ahead,
//     straight across tile boundaries (tiles are 64 rows: 64 accumulator registers beside the 256 of the weights), so the
//     next tile's pixels arrive under this tile's epilogue; one counted vmcnt and one raw barrier per stage; the tile's
//     RESIDUAL rows come by LDS-DMA too (requested when the tile starts, each wave its own 64 channels), so the epilogue
//     has no load to wait for -- with one wave per SIMD nothing else would cover an HBM latency there;
//   * workgroups that share an XCD (blockIdx mod 8) and an M lane walk the same pixel tiles with different column blocks,
//     so a pixel tile is fetched from HBM once per XCD L2 and read from L2 by the other column blocks.
// Same K order and epilogue arithmetic as conv_mfma_duo.hip / conv_mfma256.hip ((acc + bias) + residual, ReLU, round to f16):
// a layer's bits do not depend on which of the three the dispatcher picks.
#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct WsK {
    const char *x;
    const char *w;
    const float *bias;
    const char *res;
    char *y;
    int M;
    int kbytes;              // K * 2 (row pitch of x and of the packed weight rows)
    int ldy;                 // = Cout
    int relu;
    int m_tiles, n_tiles;
};

#define VKW_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void ws_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int WS_NS = 6;                     // ring slots
constexpr int WS_D = WS_NS - 1;              // stages the DMA runs ahead
constexpr int WS_BM = 64;                    // rows per tile
constexpr int WS_SLOT = WS_BM * 256;         // 64 rows x 128 channels x 2 B

// KC = K / 32 MFMA steps (4, 8, 16)
// DBG: timing-only ablation builds (VK_WS_DBG, WRONG results): 1 no epilogue, 2 no MFMA, 4 no pixel DMA, 8 no residual DMA
template <int KC, int DBG = 0>
__global__ __launch_bounds__(256, 1) void conv_ws_kernel(WsK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPT = KC / 4;              // ring stages (128 channels) per tile

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
    half8 wf[4][KC];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int co = n0 + wave * 64 + (ni >> 1) * 32 + (j >> 2) * 8 + (ni & 1) * 4 + (j & 3);
        const char *wr = p.w + (long)co * p.kbytes + g * 16;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) wf[ni][ks] = *reinterpret_cast<const half8 *>(wr + ks * 64);
    }
    char *res_lds = smem + WS_NS * WS_SLOT + wave * (WS_BM * 128);           // this wave's 64 rows x 64 channels of residual
    float *bias_lds = reinterpret_cast<float *>(smem + WS_NS * WS_SLOT + 4 * WS_BM * 128);
    bias_lds[tid] = p.bias[n0 + tid];
    ws_vm_wait<0>();                         // from here on vmcnt counts DMA pieces, residual loads and stores
    __syncthreads();

    // ---- LDS-DMA: piece q of the wave = rows (wave * 4 + q) * 4 + (lane >> 4) of the stage, 16-byte slot lane & 15 of the
    // 256-byte row, holding source chunk slot ^ (row & 15) (the swizzle sits on the source address) ----
    const int drow = lane >> 4;
    auto request = [&](int q_, int qlo = 0, int qhi = 4) {   // q_ = global stage index of this workgroup; pieces [qlo, qhi)
        const int it = q_ / SPT, st = q_ - it * SPT;
        const int m0 = (first + it * ML) * WS_BM;
        char *dst = smem + (q_ % WS_NS) * WS_SLOT + wave * 4 * 1024;
#pragma unroll
        for (int q = qlo; q < qhi; ++q) {
            const int row = (wave * 4 + q) * 4 + drow;
            const int m = min(m0 + row, p.M - 1);                             // rows past M are computed and dropped
            VKW_GLDS16(p.x + (long)m * p.kbytes + st * 256 + (((lane & 15) ^ (row & 15)) << 4), dst + q * 1024);
        }
    };
    for (int q_ = 0; q_ < WS_D && q_ < total; ++q_) request(q_);

    // fragment address of pixel tile pt, step ksl of a stage: slot + pt * 4096 + row j * 256 + swizzled chunk
    int xoff[4];
#pragma unroll
    for (int ksl = 0; ksl < 4; ++ksl) xoff[ksl] = j * 256 + (((ksl * 4 + g) ^ j) << 4);

    // ---- main loop.  The epilogue is cut into 8 units (32 channels x 16 pixels).  (Measured on this kernel: 180 us of
    // epilogue + 320 us of MFMA + 190 us of loop skeleton add up to 698 us on the 512 -> 2048 layer at M = 200 704: with one
    // wave per SIMD the phases do not overlap.  A build with two accumulator sets that issued the previous tile's units
    // between the MFMA groups was correct and no faster (K = 256: 149 vs 136 us), so one set is kept.) ----
    floatx4 acc[4][4];
    auto unit = [&](int it_, int u) {                         // unit u = (qn, pt) of tile it_'s epilogue
        const int qn = u >> 2, pt = u & 3;
        const int ch = wave * 64 + qn * 32 + g * 8;
        const int m = (first + it_ * ML) * WS_BM + pt * 16 + j;
        const floatx4 b0 = *reinterpret_cast<const floatx4 *>(bias_lds + ch), b1 = *reinterpret_cast<const floatx4 *>(bias_lds + ch + 4);
        const int row = pt * 16 + j;
        half8 rr = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.res) rr = *reinterpret_cast<const half8 *>(res_lds + row * 128 + (((qn * 4 + g) ^ (row & 7)) << 4));
        half8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x0 = (acc[pt][2 * qn][e] + b0[e]) + (float)rr[e];
            float x1 = (acc[pt][2 * qn + 1][e] + b1[e]) + (float)rr[4 + e];
            if (p.relu) {
                x0 = x0 > 0.f ? x0 : 0.f;
                x1 = x1 > 0.f ? x1 : 0.f;
            }
            o[e] = (_Float16)x0;
            o[4 + e] = (_Float16)x1;
        }
        if (m < p.M) *reinterpret_cast<half8 *>(p.y + ((long)m * p.ldy + n0 + ch) * 2) = o;
    };
    for (int it = 0; it < ntw; ++it) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[pt][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
        // residual rows of this tile -> LDS: 8 pieces of 8 rows x 128 B of this wave's 64 channels, 16-byte slot lane & 7 of
        // a row holding chunk slot ^ (row & 7)
        const int m0 = (first + it * ML) * WS_BM;
        if (p.res && !(DBG & 8)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int row = q * 8 + (lane >> 3);
                const int m = min(m0 + row, p.M - 1);
                VKW_GLDS16(p.res + ((long)m * p.ldy + n0 + wave * 64) * 2 + (((lane & 7) ^ (row & 7)) << 4), res_lds + q * 1024);
            }
        }
#pragma unroll
        for (int st = 0; st < SPT; ++st) {
            const int gs = it * SPT + st;
            if (gs + WS_D - 1 < total)
                ws_vm_wait<4 * (WS_D - 1)>();                // stage gs has landed; the later stages stay in flight
            else
                ws_vm_wait<0>();
            __builtin_amdgcn_s_barrier();                    // ... for every wave; and stage gs - 1 has been read by all
            const bool more = gs + WS_D < total && !(DBG & 4);   // a stage to request into the slot stage gs - 1 just left
            const char *slot = smem + (gs % WS_NS) * WS_SLOT;
            // all 16 fragments of the stage are requested up front; the four DMA pieces of the stage WS_D ahead are issued
            // BETWEEN the MFMA groups, where their issue cost (~100 cycles each) hides under the matrix pipe
            half8 xf[4][4];
#pragma unroll
            for (int ksl = 0; ksl < 4; ++ksl)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) xf[ksl][pt] = *reinterpret_cast<const half8 *>(slot + pt * 4096 + xoff[ksl]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ksl = 0; ksl < 4; ++ksl) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if constexpr (DBG & 2) {
                            asm volatile("" ::"v"(xf[ksl][pt]));
                            continue;
                        }
                        acc[pt][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][4 * st + ksl], xf[ksl][pt], acc[pt][ni], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (more) request(gs + WS_D, ksl, ksl + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: (acc + bias) + residual, ReLU, f16.  The residual pieces are older than the SPT stages requested since ----
        if (!(DBG & 1)) {
            if (p.res) {
                if ((it + 1) * SPT - 1 + WS_D < total)
                    ws_vm_wait<4 * SPT>();
                else
                    ws_vm_wait<0>();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) unit(it, u);
        }
    }
}

bool conv_ws_eligible(const ConvArgs &a) {
    const char *v = getenv("VK_CONV_WS");                // "0" disables (A/B switch and bit-identity tests; re-read per call)
    if (v && v[0] == '0') return false;
    if (a.stem || a.x2 || a.pool_part || a.groups > 1 || a.dt != VK_F16 || a.out_dt != VK_F16 || a.relu > 1) return false;
    if (a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride != 1) return false;
    if (a.Cout % 256 != 0 || a.ldy != a.Cout || (a.Cin != 128 && a.Cin != 256 && a.Cin != 512)) return false;
    const int nt = a.Cout / 256;
    if (32 % nt != 0) return false;                      // column blocks must tile the 32 workgroups of an XCD
    const long M = (long)a.N * a.Ho * a.Wo;
    return M >= 8 * 128 && M < (1L << 31) - 128;      // (a.Cin % 128 == 0: whole 128-channel ring stages)
}

template <int KC, int DBG = 0>
static int launch_ws(const WsK &k, hipStream_t stream) {
    constexpr int smem = WS_NS * WS_SLOT + 4 * WS_BM * 128 + 1024;
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_kernel<KC, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_ws_kernel<KC, DBG>), dim3(256), dim3(256), smem, stream, k);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_conv_ws(const ConvArgs &a, hipStream_t stream) {
    WsK k;
    k.x = (const char *)a.x;
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

    KernelTimer *tm = g_timer;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) {
        e0 = tm->get();
        e1 = tm->get();
        VK_CHECK_HIP(hipEventRecord(e0, stream));
    }
    int st;
    switch (a.Cin) {
        case 128: st = launch_ws<4>(k, stream); break;
        case 256: st = launch_ws<8>(k, stream); break;
        default: {
            const char *d = getenv("VK_WS_DBG");
            switch (d ? atoi(d) : 0) {
                case 1: st = launch_ws<16, 1>(k, stream); break;
                case 2: st = launch_ws<16, 2>(k, stream); break;
                case 3: st = launch_ws<16, 3>(k, stream); break;
                case 4: st = launch_ws<16, 4>(k, stream); break;
                case 8: st = launch_ws<16, 8>(k, stream); break;
                case 12: st = launch_ws<16, 12>(k, stream); break;
                case 15: st = launch_ws<16, 15>(k, stream); break;
                default: st = launch_ws<16>(k, stream); break;
            }
            break;
        }
    }
    VK_TRY(st);
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(e1, stream));
        tm->recs.push_back({a.concurrent ? 6 : 8, 2.0 * (double)M * a.Cout * a.Cin, e0, e1, (int)M, a.Cout, a.Cin, 1, 1,
                            2.0 * ((double)M * a.Cin + (double)M * a.Cout * (a.res ? 2 : 1) + (double)a.Cout * a.Cin)});
    }
    return VK_OK;
}

}  // namespace vk
