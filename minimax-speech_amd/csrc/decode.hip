// Decode-step projections for B <= 32 sequences (split build MMX_X3: NS = 3 planes; bf16 build: NS = 1 plane): out = epilogue( rstd * (xs W^T) ), bf16 weights
// in MFMA-fragment order (mmx_pack_skinny, no kscale), activations as PRE-SPLIT bf16 planes.  Replaces, per decode step and
// layer, the q/k/v, o, gate/up (+ SwiGLU) and down projections of HF Qwen2 as driven by
// speech/cosyvoice/llm/llm.py:359-371,745-760, and the llm_decoder head (llm.py:749).
//
// Split-plane activations.  An activation matrix x [B][K] (already multiplied by the RMSNorm gain of its consumer) is kept
// as 3 bf16 planes hi + mid + lo = x (24 significant bits), each in the A-fragment order of v_mfma_f32_16x16x32_bf16:
//     xs[plane s][m][kb][lane = g*16 + l16][j] = term s of x[m*16 + l16][kb*32 + g*8 + j]
// The PRODUCER of an activation splits it once (the epilogues below, the decode attention, mmx_decode_prep); a consumer
// workgroup loads fragments straight into MFMA operand registers: no conversion, no address arithmetic beyond one offset
// per load.  (Measured: the first version of this kernel took fp32 rows and split them per workgroup - every workgroup
// repeats the split of the whole x, ~600 VALU instructions per wave, and at batch 32 the kernel was bound by instruction
// issue, 8.7 us for the q/k/v projection, not by its 2 MB of weights.)
// The RMSNorm statistic travels the same way: a producer of the residual stream h also writes, per 16-column tile and row,
// the partial sum of squares of ITS columns (ssq[row][tile], 64 tile slots per row); the consumer adds a row's partials in
// tile order while its weight loads are in flight.
//
// Work split: a workgroup is 8 waves = 8 k slices of its K range; each wave keeps the activation fragments of its slice in
// registers for all TW output tiles of the workgroup (activation traffic per workgroup = one copy of xs, whatever TW is).
// K = 4864 (down_proj) is also cut ACROSS workgroups (gridDim.y = J): a workgroup publishes its partial tile with
// write-through stores and takes a ticket; the one whose ticket is last sums the J partials IN SLICE ORDER (the result does
// not depend on arrival order) and runs the epilogue.  Nothing ever waits or spins (a fault elsewhere cannot hang this
// kernel); the hand-off is the measured-valid form of MI355X_MICROARCH.md "Workgroup dispatch ... visibility", table row 1:
// sc1 stores, s_waitcnt vmcnt(0) in the storing wave, one agent-scope counter add by one lane of that wave, sc1 loads by
// the wave whose add returned last.
// fp16 planes (MMX_H2 / MMX_H2W, F16 = true below): the same kernels on TWO fp16 planes per activation (hi = fp16(x), lo = fp16(x - hi):
// 22 significant bits) and fp16 weights, v_mfma_f32_16x16x32_f16 (the bf16 MFMA's rate; fp16 x fp16 products are exact in fp32).
// A bf16-representable checkpoint converts to fp16 exactly (its weights are scaled by 2^8 at load so that the small ones stay
// normal numbers; the epilogue multiplies by 2^-8) and costs two MFMAs per fragment on two thirds of the activation bytes of the
// three-bf16-plane form; an fp32 checkpoint is two fp16 planes (MMX_H2W: 4 bytes per weight, three MFMAs per fragment) instead
// of three bf16 planes (MMX_X3W: 6 bytes, six MFMAs).  Range: |activation| < 65504, |weight| < 255.
#include "common.h"
#include "../../include/mmx_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
constexpr float H2_WSCALE = 256.f;                     // weights of the fp16 forms are stored as w * 2^8

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned OOB = 0x80000000u;                  // a buffer offset outside every descriptor below: the load returns 0
constexpr int SSQ_SLOTS = 64;                          // tile slots per row of a sum-of-squares table

// element index inside ONE plane of the element (row, col) of an activation with nkb k-blocks
__device__ __forceinline__ int plane_index(int row, int col, int nkb) {
    return (((((row >> 4) * nkb + (col >> 5)) << 6) + (((col & 31) >> 3) << 4) + (row & 15)) << 3) + (col & 7);
}
// v -> NS bf16 terms (NS = 3: hi + mid + lo = v; NS = 1, the bf16 build: bf16(v)) at `idx` of planes `ps` elements apart;
// F16: NS fp16 terms (round to nearest each; the remainder of a rounded fp32 value is exact in fp32)
template <int NS, bool F16 = false>
__device__ __forceinline__ void store_split(bf16_t* planes, int ps, int idx, float v) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if constexpr (F16) {
            const _Float16 h = (_Float16)v;
            planes[s * ps + idx] = __builtin_bit_cast(unsigned short, h);
            v -= (float)h;
        } else {
            const bf16_t h = f2bf(v);
            planes[s * ps + idx] = h;
            v -= bf2f(h);
        }
    }
}
template <bool F16>
__device__ __forceinline__ float4_t mma16(u32x4_t a, u32x4_t b, float4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(short8_t, a), __builtin_bit_cast(short8_t, b), c, 0, 0, 0);
}

struct Skinny3Args {
    const bf16_t* xs;        // input planes [3][MT][nkb][64][8]
    const bf16_t* wp;        // packed weights
    const float* bias;       // [N] or NULL (EPI 0 / 2)
    const float* ssq_in;     // [32][SSQ_SLOTS] partial sums of squares of the input rows, or NULL (no RMSNorm)
    float* out;              // fp32 [B][ldo]: EPI 0 result, EPI 2 residual stream (in place)
    bf16_t* xs_out;          // EPI 1: planes of silu(g)*u [3][MT][N/32][64][8]; EPI 2: planes of (out * gamma_next), or NULL
    const float* gamma_next; // EPI 2: [N] gain of the consumer of xs_out
    float* ssq_out;          // EPI 2: [32][SSQ_SLOTS] partial sums of squares of `out` rows per output tile, or NULL
    float* part;             // J > 1: partial tiles [J][tile groups][TW*MT*4][64]
    int* tickets;            // J > 1: one per tile group (workgroup column), zero between launches
    long ldo;
    int B, K, N, ntiles, kb_per_wg;
    float eps;
    unsigned wplane;         // NWP > 1 (MMX_X3W / MMX_H2W): bytes between two weight planes (each a whole pack)
    float wscale;            // the accumulator is multiplied by this (1, or 2^-8 for the fp16 forms' scaled weights)
};

// lab build only (common.h, MMX_LAB): [workgroup][wave][8] shader-clock stamps of the kernel's phases, set by mmx_lab_skinny_stamps
__device__ unsigned long long* g_skinny_stamps = nullptr;
#define STAMP(i) do { if constexpr (LAB) { if (stamps && lane == 0) stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)

// MT: 16-row tiles of the batch; TW: output tiles (EPI 1: gate/up tile pairs) per workgroup; KPW: k-blocks per wave (bound)
// NWP = 3 (MMX_X3W, an fp32 checkpoint): the weights as three bf16 planes hi + mid + lo = w, each a pack of its own; a product keeps
// the six terms (activation plane s) x (weight plane p) with s + p < 3 - both operands to fp32's 24 bits.
template <int MT, int TW, int KPW, int EPI, int NS, int NWP = 1, bool F16 = false>
__global__ __launch_bounds__(512) void skinny3_kernel(Skinny3Args a) {
    unsigned long long* stamps = nullptr;
    unsigned long long t_entry = 0;
    if constexpr (LAB) { t_entry = __builtin_amdgcn_s_memtime(); stamps = g_skinny_stamps; }
    constexpr int NB = EPI == 1 ? 2 : 1;
    constexpr int PER = TW * NB * MT * 4;              // floats per lane a wave hands to the reduction
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);       // [8 waves][PER][64 lanes]
    float* rstd = red + 8 * PER * 64;                  // [32]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, l16 = lane & 15;
    const int J = gridDim.y, j = blockIdx.y;
    // all index arithmetic in 32 bits and without divisions (a 64-bit division is ~150 scalar instructions on this ISA)
    const int nkb = a.K >> 5;
    const int wkb0 = j * a.kb_per_wg, wkb1 = min(nkb, wkb0 + a.kb_per_wg);
    const int wn = wkb1 - wkb0;
    const int kb0 = wkb0 + ((wn * wave) >> 3), kb1 = wkb0 + ((wn * (wave + 1)) >> 3);
    const int tile0 = blockIdx.x * TW;
    const int ntiles = a.ntiles;

    STAMP(0);
    if constexpr (LAB) { if (stamps && lane == 0) stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + 7] = t_entry; }
    // every load is a buffer load with a per-lane offset of lane*16 and a scalar offset; an index outside the work of this
    // wave gets the offset OOB (returns zeros: a zero fragment adds nothing) - no branch around any load
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wp), 0, 0x7fffffff, 0x00020000);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.xs), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = lane * 16;
    u32x4_t wf[NWP][TW][NB][KPW];
#pragma unroll
    for (int p2 = 0; p2 < NWP; ++p2)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int i = 0; i < KPW; ++i) {
                    const bool ok = tile0 + t < ntiles && kb0 + i < kb1;
                    const unsigned so = ok ? ((unsigned)(((tile0 + t) * NB + n) * nkb + kb0 + i) << 10) + p2 * a.wplane : OOB;
                    wf[p2][t][n][i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff, so, 2);  // aux 2: non-temporal
                }
    u32x4_t xf[NS][KPW][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned so = kb0 + i < kb1 ? ((unsigned)((s * MT + m) * nkb + kb0 + i) << 10) : OOB;
                xf[s][i][m] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, so, 0);
            }
    // RMSNorm statistic of the input rows: thread (row = tid >> 4, part = tid & 15) adds 4 of the row's tile partials, the 16
    // parts meet over 4 xor-shuffles (fixed tree), lane part 0 publishes rstd[row]; visible after the barrier below
    if (a.ssq_in) {
        const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
        const float4 q = *reinterpret_cast<const float4*>(a.ssq_in + row * SSQ_SLOTS + part * 4);
        float sq = group16_sum((q.x + q.y) + (q.z + q.w));
        if (part == 0) rstd[row] = rsqrtf(sq / (float)a.K + a.eps);
    }
    // Epilogue work is spread over all 8 waves: unit u = (tile t, row tile m, register r) covers, per lane (g, l16), the
    // element (row m*16 + 4g + r, column l16) of tile t; wave w takes units w, w + 8, ...  (One wave finishing a whole tile
    // alone - 8 values per lane through the reduction, the split into planes, the row-statistic shuffles - was the longest
    // serial stretch of the kernel.)  Operands of the epilogue are fetched up front, before the MFMA phase.
    constexpr int UNITS = TW * MT * 4, UPW = (UNITS + 7) / 8;
    float pre_bias[UPW], pre_gn[UPW], pre_res[UPW];
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = wave + 8 * k, t = u / (MT * 4), m = (u / 4) % MT, r = u & 3;
        const int row = m * 16 + 4 * g + r, ncol = (tile0 + t) * 16 + l16;
        const bool ok = u < UNITS && tile0 + t < ntiles && row < a.B && ncol < a.N;
        pre_bias[k] = (EPI != 1 && a.bias && ok) ? a.bias[ncol] : 0.f;
        pre_gn[k] = (EPI == 2 && a.gamma_next && ok) ? a.gamma_next[ncol] : 1.f;
        pre_res[k] = (EPI == 2 && ok) ? a.out[(long)row * a.ldo + ncol] : 0.f;
    }
    STAMP(1);                                          // loads issued
    if constexpr (LAB) { if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(2); } }   // (lab build only) loads landed
    float4_t acc[TW][NB][MT];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[t][n][m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int p2 = 0; p2 < NWP; ++p2) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            if (s + p2 >= NS) continue;    // (compile time) terms below the last kept bit
                            acc[t][n][m] = mma16<F16>(xf[s][i][m], wf[p2][t][n][i], acc[t][n][m]);
                        }
                }
    // reduction over the 8 k slices of the workgroup through LDS, fixed order
    {
        float* mine = red + wave * PER * 64 + lane;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[(((t * NB + n) * MT + m) * 4 + r) * 64] = acc[t][n][m][r];
    }
    STAMP(3);                                          // MFMAs done, partials in LDS
    __syncthreads();
    STAMP(4);
    float sum[UPW][NB];
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = wave + 8 * k, t = u / (MT * 4), m = (u / 4) % MT, r = u & 3;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float v = 0.f;
            if (u < UNITS) {
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) v += red[w2 * PER * 64 + (((t * NB + n) * MT + m) * 4 + r) * 64 + lane];
            }
            sum[k][n] = v;
        }
    }
    if constexpr (EPI == 2) {
        if (J > 1) {
            // cross-workgroup k split: every wave publishes its units of this slice's partial tiles (write-through), the
            // workgroup takes ONE ticket for its tile group behind a barrier (MI355X_MICROARCH.md, Guideline 16 R1: the lane that
            // signals for other waves does so behind a workgroup barrier that follows every wave's vmcnt(0)), and the
            // workgroup holding the last ticket sums the J partials in slice order
            int* flag = reinterpret_cast<int*>(rstd + 32);
#pragma unroll
            for (int k = 0; k < UPW; ++k) {
                const int u = wave + 8 * k;
                if (u < UNITS) st_sc1(a.part + ((j * gridDim.x + blockIdx.x) * UNITS + u) * 64 + lane, sum[k][0]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const int old = __hip_atomic_fetch_add(a.tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == J - 1) __hip_atomic_store(a.tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
                *flag = old == J - 1;
            }
            __syncthreads();
            STAMP(5);
            if (!*flag) return;
            // all J x UPW partial loads in flight before the first add (summed one by one behind a wait each, the last
            // arriver's epilogue was the longest phase of the kernel: 4.8 us of the down projection's 12.7)
            constexpr int JMAX = 8;
            float pv[UPW][JMAX];
#pragma unroll
            for (int k = 0; k < UPW; ++k)
#pragma unroll
                for (int j2 = 0; j2 < JMAX; ++j2) {
                    // clamped indices instead of predicated loads: no branch sits between two loads
                    const int u = min(wave + 8 * k, UNITS - 1), jc = min(j2, J - 1);
                    const float v = ld_sc1(a.part + ((jc * gridDim.x + blockIdx.x) * UNITS + u) * 64 + lane);
                    pv[k][j2] = j2 < J ? v : 0.f;
                }
#pragma unroll
            for (int k = 0; k < UPW; ++k) {
                float v = 0.f;
#pragma unroll
                for (int j2 = 0; j2 < JMAX; ++j2) v += pv[k][j2];            // slice order
                sum[k][0] = v;
            }
        }
    }
    const int nkb_out = a.N >> 5;
    const int ps_out = MT * nkb_out * 512;
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = wave + 8 * k, t = u / (MT * 4), m = (u / 4) % MT, r = u & 3;
        if (u >= UNITS) break;                         // wave-uniform
        const int row = m * 16 + 4 * g + r, tile = tile0 + t, ncol = tile * 16 + l16;
        const float sc = (a.ssq_in ? rstd[row] : 1.f) * a.wscale;
        const bool ok = tile < ntiles && row < a.B && ncol < a.N;
        if constexpr (EPI == 1) {
            const float gte = sum[k][0] * sc, up = sum[k][1] * sc;
            if (ok) store_split<NS, F16>(a.xs_out, ps_out, plane_index(row, ncol, nkb_out), gte / (1.f + expf(-gte)) * up);
        } else {
            float vv = sum[k][0] * sc + pre_bias[k];
            if constexpr (EPI == 2) vv += pre_res[k];
            if (ok) a.out[(long)row * a.ldo + ncol] = vv;
            if constexpr (EPI == 2) {
                if (ok && a.xs_out) store_split<NS, F16>(a.xs_out, ps_out, plane_index(row, ncol, nkb_out), vv * pre_gn[k]);
                if (a.ssq_out) {                       // this tile's share of the row's sum of squares: over its 16 columns
                    const float q = group16_sum(ok ? vv * vv : 0.f);
                    if (l16 == 0 && tile < ntiles) a.ssq_out[row * SSQ_SLOTS + tile] = q;
                }
            }
        }
    }
    STAMP(6);
}

template <int MT, int TW, int KPW, int EPI, int NS, int NWP = 1, bool F16 = false>
int launch(const Skinny3Args& a, int J, hipStream_t s) {
    constexpr int NB = EPI == 1 ? 2 : 1;
    const size_t lds = (size_t)8 * (TW * NB * MT * 4) * 64 * sizeof(float) + 32 * sizeof(float) + 16;
    MMX_LDS_OPT_IN((skinny3_kernel<MT, TW, KPW, EPI, NS, NWP, F16>), lds);
    dim3 grid((a.ntiles + TW - 1) / TW, J);
    hipLaunchKernelGGL((skinny3_kernel<MT, TW, KPW, EPI, NS, NWP, F16>), grid, dim3(512), lds, s, a);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// x fp32 [B][K] -> residual stream copy h, planes of (x * gamma) and the per-tile sums of squares of x: the form in which
// the first projection of a decode step wants the sampler's output (the next input embedding)
template <int NS, bool F16 = false>
__global__ __launch_bounds__(256) void decode_prep_kernel(const float* __restrict__ x, long ldx, int B, int K, const float* __restrict__ gamma,
                                                          float* __restrict__ h, long ldh, bf16_t* __restrict__ xs, int mt,
                                                          float* __restrict__ ssq) {
    const int row = blockIdx.x, nkb = K >> 5;
    const int ps = mt * nkb * 512;
    for (int t0 = threadIdx.x; t0 < SSQ_SLOTS * 16; t0 += 256) {       // thread -> (tile, column of the tile)
        const int tile = t0 >> 4, col = t0;
        float v = 0.f;
        if (col < K) {
            v = x[(long)row * ldx + col];
            if (h) h[(long)row * ldh + col] = v;
            store_split<NS, F16>(xs, ps, plane_index(row, col, nkb), v * (gamma ? gamma[col] : 1.f));
        }
        const float q = group16_sum(v * v);
        if ((t0 & 15) == 0) ssq[row * SSQ_SLOTS + tile] = q;           // tiles beyond K / 16 get 0
    }
}

}  // namespace

extern "C" int mmx_decode_prep(const float* x, int64_t ldx, int B, int K, const float* gamma, float* h, int64_t ldh, void* xs,
                               float* ssq, int dtype, hipStream_t stream) {
    MMX_CHECK_ARG(x && xs && ssq && B > 0 && B <= 32 && K > 0 && K % 32 == 0 && K <= SSQ_SLOTS * 16 && (dtype == MMX_X3 || dtype == MMX_BF16 || dtype == MMX_H2));
    if (dtype == MMX_H2) hipLaunchKernelGGL((decode_prep_kernel<2, true>), dim3(B), dim3(256), 0, stream, x, ldx, B, K, gamma, h, ldh, (bf16_t*)xs, (B + 15) / 16, ssq);
    else if (dtype == MMX_X3) hipLaunchKernelGGL(decode_prep_kernel<3>, dim3(B), dim3(256), 0, stream, x, ldx, B, K, gamma, h, ldh, (bf16_t*)xs, (B + 15) / 16, ssq);
    else hipLaunchKernelGGL(decode_prep_kernel<1>, dim3(B), dim3(256), 0, stream, x, ldx, B, K, gamma, h, ldh, (bf16_t*)xs, (B + 15) / 16, ssq);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// Shapes are those of the decode step (the template arguments bound the registers a wave needs):
//   K <= 1024 (q/k/v, o, gate/up, head): ksplit 1, <= 4 k-blocks per wave; K = 4864 (down): 8 slices of <= 3 k-blocks per wave.
extern "C" int mmx_skinny2(const void* xs, int B, int K, int N, const void* wp, const float* bias, const float* ssq_in, float eps,
                           int epi, float* out, int64_t ldo, void* xs_out, const float* gamma_next, float* ssq_out,
                           int tiles_per_wg, int ksplit, float* part, int64_t part_floats, int32_t* tickets, int dtype,
                           hipStream_t stream) {
    MMX_CHECK_ARG(xs && wp && B > 0 && B <= 32 && K > 0 && K % 32 == 0 && N > 0 &&
                  (dtype == MMX_X3 || dtype == MMX_X3W || dtype == MMX_BF16 || dtype == MMX_H2 || dtype == MMX_H2W));
    MMX_CHECK_ARG(dtype != MMX_X3W || tiles_per_wg == 1);   // three weight planes in registers: one output tile per workgroup
    MMX_CHECK_ARG(((uintptr_t)xs % 16) == 0 && ((uintptr_t)wp % 16) == 0 && (!ssq_in || ((uintptr_t)ssq_in % 16) == 0));
    MMX_CHECK_ARG(ksplit >= 1 && ksplit <= 8 && (ksplit == 1 || (epi == 2 && part && tickets)));
    MMX_CHECK_ARG(epi == 1 ? (xs_out != nullptr && N % 32 == 0) : out != nullptr);
    MMX_CHECK_ARG(epi != 2 || !xs_out || N % 32 == 0);
    MMX_CHECK_ARG(epi != 2 || !ssq_out || (N + 15) / 16 <= SSQ_SLOTS);
    const int nkb = K / 32, ntiles = (N + 15) / 16, mt = (B + 15) / 16;
    const int per_wave = ((nkb + ksplit - 1) / ksplit + 7) / 8;
    MMX_CHECK_ARG(ksplit == 1 || part_floats >= (int64_t)ksplit * ((ntiles + tiles_per_wg - 1) / tiles_per_wg) * tiles_per_wg * mt * 4 * 64);
    // 32-bit buffer offsets: the packed weights and the planes must stay below 2 GiB
    const double plane_bytes = (double)ntiles * (epi == 1 ? 2 : 1) * nkb * 1024.0;
    MMX_CHECK_ARG(plane_bytes * (dtype == MMX_X3W ? 3 : (dtype == MMX_H2W ? 2 : 1)) < 2147483000.0);
    const bool f16 = dtype == MMX_H2 || dtype == MMX_H2W;
    Skinny3Args a{(const bf16_t*)xs, (const bf16_t*)wp, bias, ssq_in, out, (bf16_t*)xs_out, gamma_next, ssq_out, part, tickets,
                  ldo, B, K, N, ntiles, (nkb + ksplit - 1) / ksplit, eps, (unsigned)plane_bytes, f16 ? 1.0f / H2_WSCALE : 1.0f};
#define GO(MT, TW, KPW, EPI) do { if (dtype == MMX_X3) return launch<MT, TW, KPW, EPI, 3>(a, ksplit, stream); \
                                  if (dtype == MMX_H2) return launch<MT, TW, KPW, EPI, 2, 1, true>(a, ksplit, stream); \
                                  if (dtype == MMX_H2W) return launch<MT, TW, KPW, EPI, 2, 2, true>(a, ksplit, stream); \
                                  if (dtype == MMX_X3W) { if constexpr (TW == 1) return launch<MT, 1, KPW, EPI, 3, 3>(a, ksplit, stream); else return MMX_EARG; } \
                                  return launch<MT, TW, KPW, EPI, 1>(a, ksplit, stream); } while (0)
#define BY_MT(TW, KPW, EPI) do { if (mt == 1) GO(1, TW, KPW, EPI); else GO(2, TW, KPW, EPI); } while (0)
    if (per_wave <= 4) {
        if (epi == 0) { if (tiles_per_wg == 1) BY_MT(1, 4, 0); if (tiles_per_wg == 2) BY_MT(2, 4, 0); }
        if (epi == 1) { if (tiles_per_wg == 1) BY_MT(1, 4, 1); if (tiles_per_wg == 2) BY_MT(2, 4, 1); }
        if (epi == 2) { if (tiles_per_wg == 1) BY_MT(1, 4, 2); if (tiles_per_wg == 2) BY_MT(2, 4, 2); }
    }
#undef BY_MT
#undef GO
    return MMX_EARG;
}

#if MMX_LAB
// lab build only (include/mmx_hip_lab.h): buf uint64 [workgroups][8 waves][8] or NULL (off); set between launches, on the current device
extern "C" int mmx_lab_skinny_stamps(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_stamps), &p, sizeof(p));
    return e == hipSuccess ? MMX_OK : -(int)e - 1000;
}
#endif
