// Decode-step projections of the split build (MMX_X3): out = epilogue( rstd * ((x * gamma) W^T) ) for B <= 32 rows of fp32
// activations against bf16 weights in MFMA-fragment order (mmx_pack_skinny, no kscale).  Replaces, per decode step and
// layer, the q/k/v, o, gate/up (+ SwiGLU) and down projections of HF Qwen2 as driven by
// speech/cosyvoice/llm/llm.py:359-371,745-760, and the llm_decoder head (llm.py:749).
//
// What the round-2 kernel (skinny_gemm_kernel, csrc/llm.hip) paid for at batch 32, and what this one does instead:
//   * every wave re-read the activation fragments of its k slice for ITS tile (x traffic = tiles x slices; at batch 32
//     the activation reads were 2 - 4 x the weight bytes of a wave).  Here a workgroup is 8 waves = 8 k slices; a wave
//     loads the fp32 activations of its slice ONCE, applies the RMSNorm gain, splits them into NS bf16 terms ONCE and
//     keeps the fragments in registers for all TW output tiles of the workgroup.  Activation traffic per workgroup = one
//     copy of x, whatever TW is;
//   * K = 4864 (down_proj) ran on 56 workgroups that each read all 622 KB of activations.  Here K is also split ACROSS
//     workgroups (gridDim.y = J slices): a workgroup reads 1 / J of x, writes its partial tile through L2 (write-through
//     stores) and takes a ticket; the workgroup whose ticket is the last one sums the J partials IN SLICE ORDER (so the
//     result does not depend on arrival order), adds bias / residual and stores.  No workgroup ever waits for another
//     (nothing spins: a fault elsewhere cannot hang this kernel), and the hand-off is the measured-valid form of
//     MI355X_MICROARCH.md "Workgroup dispatch ... visibility", table row 1: sc1 stores, s_waitcnt vmcnt(0) in the storing
//     wave, one agent-scope counter add by one lane of that wave, sc1 loads by the wave whose add returned last.
#include "common.h"
#include "../../include/mmx_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MT: 16-row tiles of the batch; NS: bf16 terms per activation; TW: output tiles (EPI 1: gate/up tile pairs) per workgroup;
// KPW: k-blocks of 32 per wave (upper bound); EPI: 0 bias, 1 SwiGLU, 2 += residual (in place)
template <int MT, int NS, int TW, int KPW, int EPI>
__global__ __launch_bounds__(512) void skinny2_kernel(const float* __restrict__ x, long ldx, int B, int K, int N,
                                                      const bf16_t* __restrict__ wp, const float* __restrict__ bias,
                                                      const float* __restrict__ kgamma, int rs, float eps,
                                                      float* __restrict__ out, long ldo, int ntiles,
                                                      float* __restrict__ part, int* __restrict__ tickets) {
    constexpr int NB = EPI == 1 ? 2 : 1;
    constexpr int PER = TW * NB * MT * 4 + MT;         // floats per lane a wave hands to the reduction
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);       // [8 waves][PER][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int J = gridDim.y, j = blockIdx.y;
    const int nkb = K / 32;
    // this workgroup's k range (slice j of J), then this wave's share of it (8 waves, as even as it gets)
    const int wkb0 = (int)((long)nkb * j / J), wkb1 = (int)((long)nkb * (j + 1) / J);
    const int wn = wkb1 - wkb0;
    const int kb0 = wkb0 + wn * wave / 8, kb1 = wkb0 + wn * (wave + 1) / 8;
    const int tile0 = blockIdx.x * TW;

    // (1) weights of every tile of the workgroup for this wave's k slice: all in flight before anything else
    u32x4_t wf[TW][NB][KPW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < KPW; ++i)
                if (tile0 + t < ntiles && kb0 + i < kb1)
                    wf[t][n][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(
                        wp + ((((long)(tile0 + t) * NB + n) * nkb + kb0 + i) * 64 + lane) * 8));
    // (2) activations of the slice (fp32, row-major; L2 resident) and the gain
    float4 xr[KPW][MT][2], gr[KPW][2];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const bool okk = kb0 + i < kb1;
        const int kk = (kb0 + i) * 32 + g * 8;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 2; ++v)
                xr[i][m][v] = (okk && m * 16 + l16 < B) ? *reinterpret_cast<const float4*>(x + (long)(m * 16 + l16) * ldx + kk + 4 * v)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int v = 0; v < 2; ++v)
            gr[i][v] = (okk && kgamma) ? *reinterpret_cast<const float4*>(kgamma + kk + 4 * v) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    // epilogue operands, fetched up front by the wave that will need them (no dependent round trip after the reduction)
    float pre_bias = 0.f, pre_res[MT][4];
    const bool epi_wave = wave < TW && tile0 + wave < ntiles;
    const int ncol = (tile0 + wave) * 16 + l16;
    if (epi_wave) {
        if (EPI != 1 && bias && ncol < N) pre_bias = bias[ncol];
        if constexpr (EPI == 2) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m * 16 + 4 * g + r;
                    pre_res[m][r] = (row < B && ncol < N) ? out[(long)row * ldo + ncol] : 0.f;
                }
        }
    }
    // (3) gain, sum of squares, split into NS bf16 terms: once per wave, reused by every tile
    short8_t xp[NS][KPW][MT];
    float ssq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ssq[m] = 0.f;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float xv[8] = {xr[i][m][0].x, xr[i][m][0].y, xr[i][m][0].z, xr[i][m][0].w, xr[i][m][1].x, xr[i][m][1].y, xr[i][m][1].z, xr[i][m][1].w};
            const float gv[8] = {gr[i][0].x, gr[i][0].y, gr[i][0].z, gr[i][0].w, gr[i][1].x, gr[i][1].y, gr[i][1].z, gr[i][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ssq[m] += xv[e] * xv[e];
                xv[e] *= gv[e];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pk[e] = pack_bf16x2(xv[2 * e], xv[2 * e + 1]);
                    xv[2 * e] -= __uint_as_float(pk[e] << 16);
                    xv[2 * e + 1] -= __uint_as_float(pk[e] & 0xffff0000u);
                }
                xp[s][i][m] = __builtin_bit_cast(short8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
            }
        }
    // (4) MFMAs
    float4_t acc[TW][NB][MT];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[t][n][m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            if (!(tile0 + t < ntiles && kb0 + i < kb1)) continue;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const short8_t bfr = __builtin_bit_cast(short8_t, wf[t][n][i]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        acc[t][n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xp[s][i][m], bfr, acc[t][n][m], 0, 0, 0);
            }
        }
    // (5) reduction over the 8 k slices of the workgroup through LDS, fixed order
    if (rs) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {                 // over the 4 k-groups of the wave: lanes l16, +16, +32, +48
            ssq[m] += __shfl_xor(ssq[m], 16, 64);
            ssq[m] += __shfl_xor(ssq[m], 32, 64);
        }
    }
    {
        float* mine = red + (long)wave * PER * 64 + lane;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[(((t * NB + n) * MT + m) * 4 + r) * 64] = acc[t][n][m][r];
#pragma unroll
        for (int m = 0; m < MT; ++m) mine[(TW * NB * MT * 4 + m) * 64] = ssq[m];
    }
    __syncthreads();
    if (!epi_wave) return;
    const int t = wave;                                // wave e finishes tile e of the workgroup
    float sum[NB][MT][4], sq[MT];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[n][m][r] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) sq[m] = 0.f;
    for (int w2 = 0; w2 < 8; ++w2) {
        const float* o = red + (long)w2 * PER * 64 + lane;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[n][m][r] += o[(((t * NB + n) * MT + m) * 4 + r) * 64];
#pragma unroll
        for (int m = 0; m < MT; ++m) sq[m] += o[(TW * NB * MT * 4 + m) * 64];
    }
    const int tile = tile0 + t;
    if constexpr (EPI == 2) {
        if (J > 1) {
            // cross-workgroup k split: publish this slice's partial tile, take a ticket, last one sums in slice order
            float* mine = part + (((long)j * ntiles + tile) * (MT * 4)) * 64 + lane;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) st_sc1(mine + (m * 4 + r) * 64, sum[0][m][r]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != J - 1) return;
            if (lane == 0) __hip_atomic_store(tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[0][m][r] = 0.f;
            for (int j2 = 0; j2 < J; ++j2) {
                const float* o = part + (((long)j2 * ntiles + tile) * (MT * 4)) * 64 + lane;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[0][m][r] += ld_sc1(o + (m * 4 + r) * 64);
            }
        }
    }
    // (6) epilogue: acc rows are 4g + r, column l16; the row's sum of squares sits on lane (row & 15) of the reduced ssq
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * 16 + 4 * g + r;
            float sc = 1.f;
            if (rs) sc = rsqrtf(__shfl(sq[m], 4 * g + r, 64) / (float)K + eps);
            if (row >= B || ncol >= N) continue;
            if constexpr (EPI == 1) {
                const float gte = sum[0][m][r] * sc, up = sum[1][m][r] * sc;
                out[(long)row * ldo + ncol] = gte / (1.f + expf(-gte)) * up;
            } else {
                float vv = sum[0][m][r] * sc + pre_bias;
                if constexpr (EPI == 2) vv += pre_res[m][r];
                out[(long)row * ldo + ncol] = vv;
            }
        }
}

template <int MT, int NS, int TW, int KPW, int EPI>
int launch(const float* x, long ldx, int B, int K, int N, const bf16_t* wp, const float* bias, const float* kgamma, int rs, float eps,
           float* out, long ldo, int J, float* part, int* tickets, hipStream_t s) {
    constexpr int NB = EPI == 1 ? 2 : 1;
    const int ntiles = (N + 15) / 16;
    const size_t lds = (size_t)8 * (TW * NB * MT * 4 + MT) * 64 * sizeof(float);
    MMX_CHECK_ARG(lds <= 160 * 1024);
    MMX_LDS_OPT_IN((skinny2_kernel<MT, NS, TW, KPW, EPI>), lds);
    dim3 grid((ntiles + TW - 1) / TW, J);
    hipLaunchKernelGGL((skinny2_kernel<MT, NS, TW, KPW, EPI>), grid, dim3(512), lds, s, x, ldx, B, K, N, wp, bias, kgamma, rs, eps, out, ldo,
                       ntiles, part, tickets);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

}  // namespace

// Shapes are those of the decode step (the template arguments bound the registers a wave needs):
//   K <= 1024 (q/k/v, o, gate/up, head): J = 1, <= 4 k-blocks per wave; K = 4864 (down): J = 8 slices of <= 3 k-blocks per wave.
extern "C" int mmx_skinny2(const float* x, int64_t ldx, int B, int K, int N, const void* wp, const float* bias, const float* kgamma,
                           int rs, float eps, int epi, float* out, int64_t ldo, int tiles_per_wg, int ksplit, float* part,
                           int64_t part_floats, int32_t* tickets, int dtype, hipStream_t stream) {
    MMX_CHECK_ARG(x && wp && out && B > 0 && B <= 32 && K > 0 && K % 32 == 0 && N > 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0);
    MMX_CHECK_ARG(((uintptr_t)wp % 16) == 0 && (!kgamma || ((uintptr_t)kgamma % 16) == 0) && dtype == MMX_X3);
    MMX_CHECK_ARG(ksplit >= 1 && (ksplit == 1 || (epi == 2 && part && tickets)));
    const int nkb = K / 32, ntiles = (N + 15) / 16, mt = (B + 15) / 16;
    const int per_wave = ((nkb + ksplit - 1) / ksplit + 7) / 8;
    MMX_CHECK_ARG(ksplit == 1 || part_floats >= (int64_t)ksplit * ntiles * mt * 4 * 64);
    const bf16_t* w = (const bf16_t*)wp;
#define GO(MT, NS, TW, KPW, EPI) return launch<MT, NS, TW, KPW, EPI>(x, ldx, B, K, N, w, bias, kgamma, rs, eps, out, ldo, ksplit, part, tickets, stream)
#define BY_NS(MT, TW, KPW, EPI) do { GO(MT, 3, TW, KPW, EPI); } while (0)
#define BY_MT(TW, KPW, EPI) do { if (mt == 1) BY_NS(1, TW, KPW, EPI); else BY_NS(2, TW, KPW, EPI); } while (0)
    if (epi == 0 && per_wave <= 4) { if (tiles_per_wg == 1) BY_MT(1, 4, 0); if (tiles_per_wg == 2) BY_MT(2, 4, 0); }
    if (epi == 1 && per_wave <= 4) { if (tiles_per_wg == 1) BY_MT(1, 4, 1); if (tiles_per_wg == 2) BY_MT(2, 4, 1); }
    if (epi == 2 && per_wave <= 4) { if (tiles_per_wg == 1) BY_MT(1, 4, 2); if (tiles_per_wg == 2) BY_MT(2, 4, 2); }
#undef BY_MT
#undef BY_NS
#undef GO
    return MMX_EARG;
}
