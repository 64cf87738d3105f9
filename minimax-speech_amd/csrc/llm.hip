// Autoregressive speech-token LM decode step (Qwen2-0.5B backbone) for gfx950.
// At batch 1..64 every projection is a weight-streaming problem (727.6 MB of bf16 weights per token,
// SURVEY.md §8d): weights are pre-packed in MFMA-fragment order so one wave-instruction reads a
// contiguous 1 KiB straight into registers (no LDS round trip: cdna_hip_programming.md §5 "GEMV / M <= 16"),
// the <= 64 activation rows ride the A operand of v_mfma_f32_16x16x32_bf16, K is split across the waves
// of a workgroup and reduced through LDS (deterministic: no atomics), RMSNorm is folded into the
// projection (its weight into Wp at pack time, 1/rms as a per-row scale computed from the same x loads).
#include "common.h"
#include "../../include/mmx_hip.h"

// ------------------------------------------------------------------------------------------ pack
// bf16: Wp[tile][kt][lane][8]  = W[row(tile, lane&15)][kt*32 + 8*(lane>>4) + j] * kscale[k]
// fp32: Wp[tile][kc][lane][4]  = W[row(tile, lane&15)][kc*16 + 4*(lane>>4) + s] * kscale[k]
// interleave_half (SwiGLU): N = 2I; packed tile 2t = gate rows 16t.., 2t+1 = up rows I+16t..
template <typename T>
__global__ void pack_skinny_kernel(const T* __restrict__ w, long ldw, int N, int K, const float* __restrict__ kscale,
                                   int half, T* __restrict__ wp) {
    constexpr int E = sizeof(T) == 2 ? 8 : 4;          // elements per lane per k block
    constexpr int KB = E * 4;                          // k per block (32 / 16)
    const long total = (long)((N + 15) / 16) * (K / KB) * 64 * E;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int j = idx % E;
        long t = idx / E;
        int lane = t % 64;
        t /= 64;
        int kb = t % (K / KB);
        int tile = t / (K / KB);
        int col = lane & 15, g = lane >> 4;
        int row;
        if (half > 0) row = (tile & 1) * half + (tile >> 1) * 16 + col;
        else row = tile * 16 + col;
        int kk = kb * KB + g * E + j;
        float v = 0.f;
        if (row < N && (half == 0 || ((tile >> 1) * 16 + col) < half)) {
            v = Cvt<T>::to_f(w[(long)row * ldw + kk]);
            if (kscale) v *= kscale[kk];
        }
        wp[idx] = Cvt<T>::from_f(v);
    }
}
extern "C" int mmx_pack_skinny(const void* w, int64_t ldw, int N, int K, const float* kscale, int interleave_half,
                               void* wp, int dtype, hipStream_t stream) {
    MMX_CHECK_ARG(w && wp && N > 0 && K > 0 && K % 32 == 0);
    MMX_CHECK_ARG(interleave_half == 0 || (N == 2 * interleave_half && interleave_half % 16 == 0));
    if (dtype == MMX_X2 || dtype == MMX_X3) { MMX_CHECK_ARG(!kscale); dtype = MMX_BF16; }   // split build: exact bf16 weights
    if (dtype == MMX_BF16) hipLaunchKernelGGL(pack_skinny_kernel<bf16_t>, dim3(2048), dim3(256), 0, stream, (const bf16_t*)w, ldw, N, K, kscale, interleave_half, (bf16_t*)wp);
    else if (dtype == MMX_F32) hipLaunchKernelGGL(pack_skinny_kernel<float>, dim3(2048), dim3(256), 0, stream, (const float*)w, ldw, N, K, kscale, interleave_half, (float*)wp);
    else return MMX_EARG;
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// ------------------------------------------------------------------------------------------ packed activations
// At batch > 8 the <= 64 activation rows are kept in the A-fragment order of the MFMA as well,
//   xp[m][kb][lane][E],  lane = g*16 + l16  <->  x[m*16 + l16][kb*KB + g*E + j]   (E = 8 bf16 / 4 fp32, KB = 4E),
// so that a fragment load is one contiguous 1 KiB per wave-instruction instead of 16 row segments of 64 B: every
// workgroup re-reads the whole activation matrix from L2, and at batch 32 those reads cost more than the weight
// stream (tools/skinny_lab.hip: 10.6 -> 8.6 us gate/up, 15.2 -> 10.7 us down).  Producers (the GEMM epilogues
// and the decode attention) write this layout directly.  At batch <= 8 row-major wins (only the valid rows move).
template <typename T>
__device__ __forceinline__ long act_packed_index(int row, int col, int K) {
    constexpr int E = sizeof(T) == 2 ? 8 : 4, KB = 4 * E;
    const int kb = col / KB, g = (col % KB) / E, j = col % E;
    return ((((long)(row >> 4) * (K / KB) + kb) * 64) + g * 16 + (row & 15)) * E + j;
}

// ------------------------------------------------------------------------------------------ skinny GEMM
// wave -> (output tile of 16 columns [x2 for SwiGLU], k slice); MT = row tiles of 16 (B <= 16*MT).
// The whole problem is latency bound (a projection is 1.6 - 17 MB, a single HBM round trip is ~1-2 us), so
// bandwidth comes from bytes in flight: every wave issues ALL the weight loads of its k slice (<= KS k-blocks,
// 1 KiB per wave-instruction) back to back into registers before the first MFMA, and the k split is chosen so
// that a slice fits (skinny_launch_mt below).
// NS > 1 (MMX_X2 / MMX_X3): bf16 weights, fp32 activations split into NS bf16 terms per product (see csrc/gemm.hip);
// the RMSNorm gain is then NOT folded into the weights (they must stay the checkpoint's bf16 values): kgamma [K]
// multiplies the activations before the split.
template <typename T, typename TX, int MT, int EPI, int KS, bool XPK, bool OPK, int NS = 1>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const TX* __restrict__ x, long ldx, int B, int K, int N,
                                   const T* __restrict__ wp, const float* __restrict__ bias, int rs, float eps,
                                   float* __restrict__ outf, long ldo_f, T* __restrict__ outa, long ldo_a,
                                   int ksplit, int ntiles, const float* __restrict__ kgamma) {
    constexpr bool BF = sizeof(T) == 2;
    static_assert(NS == 1 || (BF && sizeof(TX) == 4 && !XPK && !OPK), "split build: bf16 weights, fp32 row-major activations");
    constexpr int E = BF ? 8 : 4;
    constexpr int KB = E * 4;
    constexpr int NB = EPI == 1 ? 2 : 1;               // B fragments per wave
    constexpr int XV = sizeof(TX) == 4 ? E / 4 : 1;    // 16-byte loads per A fragment
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);       // [waves][NB*MT*4 + MT] x 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int tpb = nwaves / ksplit;
    const int tile = blockIdx.x * tpb + wave / ksplit;
    const int ksl = wave % ksplit;
    const int nkb = K / KB;
    const int kb_per = (nkb + ksplit - 1) / ksplit;
    const int kb0 = ksl * kb_per, kb1 = min(nkb, kb0 + kb_per);
    const bool active = tile < ntiles;

    float4_t acc[NB][MT];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = float4_t{0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ssq[m] = 0.f;
    constexpr bool SSQ_MFMA = BF && sizeof(TX) == 2 && EPI != 2;     // EPI 2 (residual projections) never normalises
    float4_t accd[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) accd[m] = float4_t{0.f, 0.f, 0.f, 0.f};
    // epilogue operands are fetched up front (by the wave that will run the epilogue) so they do not add a
    // dependent memory round trip after the reduction
    float pre_bias = 0.f, pre_res[MT][4];
    if (active && ksl == 0) {
        const int n = tile * 16 + l16;
        if (EPI != 1 && bias && n < N) pre_bias = bias[n];
        if constexpr (EPI == 2) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m * 16 + 4 * g + r;
                    pre_res[m][r] = (row < B && n < N) ? outf[(long)row * ldo_f + n] : 0.f;
                }
        }
    }

    if (active) {
        const T* wbase = wp + ((long)tile * NB * nkb) * 64 * E + (long)lane * E;
        for (int kc = kb0; kc < kb1; kc += KS) {
            // (1) the whole weight slice in flight
            u32x4_t wf[KS][NB];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    if (kc + i < kb1)
                        wf[i][n] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wbase + ((long)n * nkb + kc + i) * 64 * E));
            // (2) activations (L2 resident) of the slice, also all in flight
            u32x4_t xr[KS][MT][XV];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int v = 0; v < XV; ++v) {
                        if constexpr (XPK)                       // packed activations: 1 KiB contiguous per fragment
                            xr[i][m][v] = (kc + i < kb1)
                                              ? *reinterpret_cast<const u32x4_t*>(x + (((long)m * nkb + kc + i) * 64 + lane) * E)
                                              : u32x4_t{0, 0, 0, 0};
                        else
                            xr[i][m][v] = (kc + i < kb1 && m * 16 + l16 < B)
                                              ? *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(x + (long)(m * 16 + l16) * ldx + (kc + i) * KB + g * E) + v * 16)
                                              : u32x4_t{0, 0, 0, 0};
                    }
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                if (kc + i >= kb1) continue;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float xv[E];
                    u32x4_t raw[XV];
#pragma unroll
                    for (int v = 0; v < XV; ++v) raw[v] = xr[i][m][v];
                    if constexpr (sizeof(TX) == 4) {
#pragma unroll
                        for (int e = 0; e < E; ++e) xv[e] = __uint_as_float(raw[e / 4][e % 4]);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) xv[e] = __uint_as_float((raw[0][e / 2] >> ((e & 1) * 16)) << 16);
                    }
                    if constexpr (SSQ_MFMA) {
                        // sum of squares on the matrix pipe: the fragment times itself is X X^T, whose diagonal is the
                        // per-row sum of squares (bf16 products are exact in fp32); 16 VALU ops per fragment otherwise
                        if (rs) {
                            const short8_t xf = *reinterpret_cast<const short8_t*>(&raw[0]);
                            accd[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, xf, accd[m], 0, 0, 0);
                        }
                    } else if (rs) {
#pragma unroll
                        for (int e = 0; e < E; ++e) ssq[m] += xv[e] * xv[e];
                    }
                    if constexpr (NS > 1 || (BF && sizeof(TX) == 4)) {         // fp32 activations: gain, then NS bf16 terms
                        if (kgamma) {
                            const float4 g0 = *reinterpret_cast<const float4*>(kgamma + (kc + i) * KB + g * E);
                            const float4 g1 = *reinterpret_cast<const float4*>(kgamma + (kc + i) * KB + g * E + 4);
                            xv[0] *= g0.x; xv[1] *= g0.y; xv[2] *= g0.z; xv[3] *= g0.w;
                            xv[4] *= g1.x; xv[5] *= g1.y; xv[6] *= g1.z; xv[7] *= g1.w;
                        }
#pragma unroll
                        for (int s2 = 0; s2 < NS; ++s2) {
                            short8_t af;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const bf16_t hb = f2bf(xv[e]);
                                af[e] = (short)hb;
                                xv[e] -= bf2f(hb);
                            }
#pragma unroll
                            for (int n = 0; n < NB; ++n) {
                                short8_t bfr = *reinterpret_cast<const short8_t*>(&wf[i][n]);
                                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[n][m], 0, 0, 0);
                            }
                        }
                    } else if constexpr (BF) {
                        short8_t af;
                        if constexpr (sizeof(TX) == 2) {
                            af = *reinterpret_cast<const short8_t*>(&raw[0]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) af[e] = (short)f2bf(xv[e]);
                        }
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            short8_t bfr = *reinterpret_cast<const short8_t*>(&wf[i][n]);
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[n][m], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            float4_t bfr = *reinterpret_cast<const float4_t*>(&wf[i][n]);
#pragma unroll
                            for (int s2 = 0; s2 < 4; ++s2)
                                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s2], bfr[s2], acc[n][m], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (SSQ_MFMA) {
        // C layout: lane (g, l16) holds rows 4g..4g+3 of column l16; the diagonal element of row l16 sits on the lane
        // with g == l16 / 4, register l16 % 4.  The butterfly below then spreads it to the other three k-groups.
        if (rs) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int r = l16 & 3;
                const float d = r == 0 ? accd[m][0] : (r == 1 ? accd[m][1] : (r == 2 ? accd[m][2] : accd[m][3]));
                ssq[m] = (l16 >> 2) == g ? d : 0.f;
            }
        }
    }
    // sum of squares: reduce across the 4 k-groups of the wave (lanes l16, l16+16, +32, +48)
    if (rs) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ssq[m] += __shfl_xor(ssq[m], 16, 64);
            ssq[m] += __shfl_xor(ssq[m], 32, 64);
        }
    }
    // cross-wave (k split) reduction through LDS, fixed order -> deterministic
    constexpr int PER = NB * MT * 4 + MT;
    if (ksplit > 1) {
        float* mine = red + ((long)wave * PER) * 64 + lane;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[((n * MT + m) * 4 + r) * 64] = acc[n][m][r];
#pragma unroll
        for (int m = 0; m < MT; ++m) mine[(NB * MT * 4 + m) * 64] = ssq[m];
        __syncthreads();
        if (ksl != 0) return;
        for (int w2 = 1; w2 < ksplit; ++w2) {
            const float* o = red + ((long)(wave + w2) * PER) * 64 + lane;
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[n][m][r] += o[((n * MT + m) * 4 + r) * 64];
#pragma unroll
            for (int m = 0; m < MT; ++m) ssq[m] += o[(NB * MT * 4 + m) * 64];
        }
    }
    if (!active) return;
    // epilogue: acc rows are 4g + r, columns l16; ssq is held per A-row l16 -> fetch row 4g+r from lane 4g+r
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m * 16 + 4 * g + r;
            float sc = 1.f;
            if (rs) {
                float sq = __shfl(ssq[m], 4 * g + r, 64);
                sc = rsqrtf(sq / (float)K + eps);
            }
            if (row >= B) continue;
            if constexpr (EPI == 1) {
                const int n = tile * 16 + l16;
                if (n < N) {
                    float gte = acc[0][m][r] * sc, up = acc[1][m][r] * sc;
                    float sl = gte / (1.f + expf(-gte));
                    if constexpr (NS > 1) outf[(long)row * ldo_f + n] = sl * up;       // split build: activations are fp32
                    else outa[OPK ? act_packed_index<T>(row, n, N) : (long)row * ldo_a + n] = Cvt<T>::from_f(sl * up);
                }
            } else {
                const int n = tile * 16 + l16;
                if (n < N) {
                    float v = acc[0][m][r] * sc + pre_bias;
                    if constexpr (EPI == 2) v += pre_res[m][r];
                    if (outf) outf[(long)row * ldo_f + n] = v;
                    if constexpr (NS == 1)
                        if (outa) outa[OPK ? act_packed_index<T>(row, n, N) : (long)row * ldo_a + n] = Cvt<T>::from_f(v);
                }
            }
        }
    }
}

template <typename T, typename TX, int EPI, bool XPK, bool OPK, int NS = 1>
static int skinny_launch_mt(const void* x, int64_t ldx, int B, int K, int N, const void* wp, const float* bias, int rs,
                            float eps, float* outf, int64_t ldo_f, void* outa, int64_t ldo_a, hipStream_t s,
                            const float* kgamma = nullptr) {
    constexpr int KB = sizeof(T) == 2 ? 32 : 16;
    const int ntiles = (N + 15) / 16;                  // for EPI==1, N is the activation width I
    const int nkb = K / KB;
    // smallest power-of-two k split (<= 8 waves = 512 threads per workgroup, so a wave may use 256 VGPRs) whose
    // slice fits KS blocks; a longer slice is walked in chunks of KS
    const int mt = (B + 15) / 16;
    const int KSr = mt == 1 ? 10 : (mt == 2 ? 7 : 4);  // k-blocks a wave keeps in flight (register budget)
    int ksplit = 1;
    while (ksplit < 8 && (nkb + ksplit - 1) / ksplit > KSr) ksplit *= 2;
    int waves = ksplit >= 4 ? ksplit : 4;
    // batch > 16 and a wide N (gate/up): two tiles per workgroup - the waves of the same k slice then share their
    // activation fragments through the L1 (tools/skinny_lab.hip: 9.06 -> 8.55 us)
    if (mt >= 2 && ksplit <= 4 && ntiles >= 256) waves = 8;
    int tpb = waves / ksplit;
    dim3 grid((ntiles + tpb - 1) / tpb), block(waves * 64);
    constexpr int NB = EPI == 1 ? 2 : 1;
    size_t lds = ksplit > 1 ? (size_t)waves * (NB * mt * 4 + mt) * 64 * 4 : 0;
#define SK(MT) hipLaunchKernelGGL((skinny_gemm_kernel<T, TX, MT, EPI, (MT == 1 ? 10 : (MT == 2 ? 7 : 4)), XPK, OPK, NS>), grid, block, lds, s, (const TX*)x, ldx, B, K, N, (const T*)wp, \
        bias, rs, eps, outf, ldo_f, (T*)outa, ldo_a, ksplit, ntiles, kgamma)
    switch (mt) {
        case 1: SK(1); break;
        case 2: SK(2); break;
        case 3: SK(3); break;
        case 4: SK(4); break;
        default: return MMX_EARG;
    }
#undef SK
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
template <typename T, typename TX, bool XPK, bool OPK>
static int skinny_launch_epi(int epi, const void* x, int64_t ldx, int B, int K, int N, const void* wp, const float* bias,
                             int rs, float eps, float* outf, int64_t ldo_f, void* outa, int64_t ldo_a, hipStream_t s) {
    if (epi == 0) return skinny_launch_mt<T, TX, 0, XPK, OPK>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
    if (epi == 1) return skinny_launch_mt<T, TX, 1, XPK, OPK>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
    if (epi == 2) return skinny_launch_mt<T, TX, 2, XPK, OPK>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
    return MMX_EARG;
}
// the activation layouts are compile-time (a runtime switch inside the unrolled load block cost ~1 us per launch at
// batch 1); the combinations the decode step uses: row-major, packed x only, packed x and packed out_act
template <typename T, typename TX>
static int skinny_launch_layout(int flags, int epi, const void* x, int64_t ldx, int B, int K, int N, const void* wp,
                                const float* bias, int rs, float eps, float* outf, int64_t ldo_f, void* outa, int64_t ldo_a,
                                hipStream_t s) {
    if constexpr (sizeof(T) == sizeof(TX)) {
        if (flags == (MMX_X_PACKED | MMX_OUT_PACKED))
            return skinny_launch_epi<T, TX, true, true>(epi, x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
        if (flags == MMX_X_PACKED)
            return skinny_launch_epi<T, TX, true, false>(epi, x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
    }
    if (flags == 0) return skinny_launch_epi<T, TX, false, false>(epi, x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, outa, ldo_a, s);
    return MMX_EARG;
}
template <int NS>
static int skinny_launch_split(int epi, const void* x, int64_t ldx, int B, int K, int N, const void* wp, const float* bias, int rs,
                               float eps, float* outf, int64_t ldo_f, hipStream_t s, const float* kgamma) {
    if (epi == 0) return skinny_launch_mt<bf16_t, float, 0, false, false, NS>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, nullptr, 0, s, kgamma);
    if (epi == 1) return skinny_launch_mt<bf16_t, float, 1, false, false, NS>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, nullptr, 0, s, kgamma);
    if (epi == 2) return skinny_launch_mt<bf16_t, float, 2, false, false, NS>(x, ldx, B, K, N, wp, bias, rs, eps, outf, ldo_f, nullptr, 0, s, kgamma);
    return MMX_EARG;
}
extern "C" int mmx_skinny_gemm(const void* x, int x_dtype, int64_t ldx, int B, int K, int N, const void* wp,
                               const float* bias, int rs, float eps, int epi, float* out_f32, int64_t ldo_f,
                               void* out_act, int64_t ldo_a, int dtype, int flags, const float* kgamma, hipStream_t stream) {
    MMX_CHECK_ARG(x && wp && B > 0 && B <= 64 && K > 0 && K % 32 == 0 && N > 0);
    if (dtype == MMX_X2 || dtype == MMX_X3) {
        // split build: fp32 row-major activations in, fp32 out (epi 1 writes SwiGLU(gate, up) to out_f32)
        MMX_CHECK_ARG(x_dtype == MMX_F32 && flags == 0 && out_f32 && !out_act && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wp % 16) == 0);
        MMX_CHECK_ARG(!kgamma || ((uintptr_t)kgamma % 16) == 0);
        return dtype == MMX_X2 ? skinny_launch_split<2>(epi, x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, stream, kgamma)
                               : skinny_launch_split<3>(epi, x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, stream, kgamma);
    }
    if (kgamma) {
        // bf16 build with the gain applied to fp32 activations in the kernel (weights packed WITHOUT kscale: the decode step
        // of that build runs on csrc/decode.hip, whose producers apply the gain; this kernel serves its prompt chunks)
        MMX_CHECK_ARG(dtype == MMX_BF16 && x_dtype == MMX_F32 && flags == 0 && ((uintptr_t)kgamma % 16) == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0);
        if (epi == 0) return skinny_launch_mt<bf16_t, float, 0, false, false, 1>(x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, out_act, ldo_a, stream, kgamma);
        if (epi == 1) return skinny_launch_mt<bf16_t, float, 1, false, false, 1>(x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, out_act, ldo_a, stream, kgamma);
        return MMX_EARG;
    }
    MMX_CHECK_ARG((flags == 0 || flags == MMX_X_PACKED || flags == (MMX_X_PACKED | MMX_OUT_PACKED)) && (flags == 0 || x_dtype == dtype) &&
                  (!(flags & MMX_OUT_PACKED) || N % 32 == 0));
    MMX_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wp % 16) == 0);
    MMX_CHECK_ARG(epi == 1 ? out_act != nullptr : (out_f32 != nullptr || (epi == 0 && out_act != nullptr)));
    MMX_CHECK_ARG(epi != 2 || out_f32 != nullptr);
    if (dtype == MMX_BF16) {
        if (x_dtype == MMX_F32) return skinny_launch_layout<bf16_t, float>(flags, epi, x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, out_act, ldo_a, stream);
        if (x_dtype == MMX_BF16) return skinny_launch_layout<bf16_t, bf16_t>(flags, epi, x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, out_act, ldo_a, stream);
    } else if (dtype == MMX_F32 && x_dtype == MMX_F32) {
        return skinny_launch_layout<float, float>(flags, epi, x, ldx, B, K, N, wp, bias, rs, eps, out_f32, ldo_f, out_act, ldo_a, stream);
    }
    return MMX_EARG;
}

// ------------------------------------------------------------------------------------------ RoPE + paged KV append
template <typename T>
__global__ void rope_kv_kernel(const float* __restrict__ qkv, long ldqkv, long qkv_bs, int Hq, int Hkv,
                               const float* __restrict__ inv_freq, const int32_t* __restrict__ pos,
                               T* __restrict__ q_out, long ldq, long q_bs, T* __restrict__ kc, T* __restrict__ vc,
                               const int32_t* __restrict__ block_table, int max_pages, int page) {
    constexpr int D = 64, HALF = 32;
    const int b = blockIdx.y, t = blockIdx.x;
    const int p = pos[b] + t;
    const float* src = qkv + (long)b * qkv_bs + (long)t * ldqkv;
    const int nh = Hq + 2 * Hkv;
    const int phys = block_table[(long)b * max_pages + p / page];
    const int slot = p % page;
    for (int i = threadIdx.x; i < nh * HALF; i += blockDim.x) {
        const int hh = i / HALF, d = i % HALF;
        const float x0 = src[hh * D + d], x1 = src[hh * D + d + HALF];
        float y0 = x0, y1 = x1;
        if (hh < Hq + Hkv) {                            // q and k heads are rotated, v is not
            const float ang = (float)p * inv_freq[d];
            const float c = cosf(ang), s = sinf(ang);
            y0 = x0 * c - x1 * s;
            y1 = x1 * c + x0 * s;
        }
        if (hh < Hq) {
            T* o = q_out + (long)b * q_bs + (long)t * ldq + hh * D;
            o[d] = Cvt<T>::from_f(y0);
            o[d + HALF] = Cvt<T>::from_f(y1);
        } else {
            const int hk = (hh - Hq) % Hkv;
            T* base = (hh < Hq + Hkv ? kc : vc) + (((long)phys * Hkv + hk) * page + slot) * D;
            base[d] = Cvt<T>::from_f(y0);
            base[d + HALF] = Cvt<T>::from_f(y1);
        }
    }
}
extern "C" int mmx_rope_kv_store(const float* qkv, int64_t ldqkv, int64_t qkv_bs, int B, int rows, int Hq, int Hkv, int D,
                                 const float* inv_freq, const int32_t* pos, void* q_out, int64_t ldq, int64_t q_bs,
                                 void* kc, void* vc, const int32_t* block_table, int max_pages, int page,
                                 int dtype, hipStream_t stream) {
    dtype = MMX_ACT_DTYPE(dtype);
    MMX_CHECK_ARG(qkv && inv_freq && pos && q_out && kc && vc && block_table && B > 0 && rows > 0 && D == 64 && page > 0);
    dim3 grid(rows, B);
    if (dtype == MMX_BF16) hipLaunchKernelGGL(rope_kv_kernel<bf16_t>, grid, dim3(256), 0, stream, qkv, ldqkv, qkv_bs, Hq, Hkv, inv_freq, pos, (bf16_t*)q_out, ldq, q_bs, (bf16_t*)kc, (bf16_t*)vc, block_table, max_pages, page);
    else if (dtype == MMX_F32) hipLaunchKernelGGL(rope_kv_kernel<float>, grid, dim3(256), 0, stream, qkv, ldqkv, qkv_bs, Hq, Hkv, inv_freq, pos, (float*)q_out, ldq, q_bs, (float*)kc, (float*)vc, block_table, max_pages, page);
    else return MMX_EARG;
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// ------------------------------------------------------------------------------------------ paged causal GQA attention
template <typename T>
__global__ __launch_bounds__(256) void paged_attn_kernel(
    const T* __restrict__ q, long ldq, long q_bs, int Hq, int Hkv, float scale, const int32_t* __restrict__ pos,
    const T* __restrict__ kc, const T* __restrict__ vc, const int32_t* __restrict__ block_table, int max_pages,
    int page, T* __restrict__ out, long ldo, long o_bs) {
    constexpr int D = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);        // [D]
    float* red = qs + D;                               // [8 + 4*D]
    float* S = red + 8 + 4 * D;                        // [ctx]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
    const int hk = h / (Hq / Hkv);
    const int ctx = pos[b] + t + 1;
    const int32_t* bt = block_table + (long)b * max_pages;
    if (tid < D) qs[tid] = Cvt<T>::to_f(q[(long)b * q_bs + (long)t * ldq + h * D + tid]);
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < ctx; j += 256) {
        const T* kp = kc + (((long)bt[j / page] * Hkv + hk) * page + j % page) * D;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s += qs[d] * Cvt<T>::to_f(kp[d]);
        s *= scale;
        S[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int j = tid; j < ctx; j += 256) {
        float e = expf(S[j] - mx);
        S[j] = e;
        l += e;
    }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];
    // PV: thread -> (d = lane, key partition = wave)
    float acc = 0.f;
    for (int j = wave; j < ctx; j += 4) {
        const T* vp = vc + (((long)bt[j / page] * Hkv + hk) * page + j % page) * D;
        acc += S[j] * Cvt<T>::to_f(vp[lane]);
    }
    red[8 + wave * D + lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float o = (red[8 + lane] + red[8 + D + lane]) + (red[8 + 2 * D + lane] + red[8 + 3 * D + lane]);
        out[(long)b * o_bs + (long)t * ldo + h * D + lane] = Cvt<T>::from_f(o / l);
    }
}
extern "C" int mmx_paged_attn(const void* q, int64_t ldq, int64_t q_bs, int B, int rows, int Hq, int Hkv, int D, float scale,
                              const int32_t* pos, const void* kc, const void* vc, const int32_t* block_table, int max_pages,
                              int page, void* out, int64_t ldo, int64_t o_bs, int dtype, hipStream_t stream) {
    dtype = MMX_ACT_DTYPE(dtype);
    MMX_CHECK_ARG(q && pos && kc && vc && block_table && out && B > 0 && rows > 0 && D == 64 && Hq % Hkv == 0 && page > 0);
    const size_t max_ctx = (size_t)max_pages * page;
    size_t lds = (64 + 8 + 4 * 64 + max_ctx) * 4;
    MMX_CHECK_ARG(lds <= 160 * 1024);
    dim3 grid(Hq, rows, B);
    if (dtype == MMX_BF16) hipLaunchKernelGGL(paged_attn_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)q, ldq, q_bs, Hq, Hkv, scale, pos, (const bf16_t*)kc, (const bf16_t*)vc, block_table, max_pages, page, (bf16_t*)out, ldo, o_bs);
    else if (dtype == MMX_F32) hipLaunchKernelGGL(paged_attn_kernel<float>, grid, dim3(256), lds, stream, (const float*)q, ldq, q_bs, Hq, Hkv, scale, pos, (const float*)kc, (const float*)vc, block_table, max_pages, page, (float*)out, ldo, o_bs);
    else return MMX_EARG;
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// ------------------------------------------------------------------------------------------ fused decode attention
// One launch per layer for the single-token step: RoPE(q), RoPE(k_new), KV append and causal GQA attention.
// Every (head, sequence) workgroup ropes the new key itself (so it does not wait for the cache write of the
// one workgroup per kv head that appends it), reads the older keys/values from the paged cache with 16-byte
// loads all in flight, and splits PV over 32 key groups x 8 channel chunks.
template <typename T>
__device__ __forceinline__ void load8(const T* p, float o[8]) {
    if constexpr (sizeof(T) == 2) {
        uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    } else {
        float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}
// 8 consecutive elements kept RAW in registers (16 B for bf16, 32 B for fp32): a thread holds the K and V chunks of
// many keys in flight and only converts a chunk when it consumes it
template <typename T>
struct Raw8 {
    uint4 a, b;                                        // b is unused for 2-byte T (the compiler drops it)
    __device__ __forceinline__ void load(const T* p) {
        a = *reinterpret_cast<const uint4*>(p);
        if constexpr (sizeof(T) == 4) b = *reinterpret_cast<const uint4*>(p + 4);
    }
    __device__ __forceinline__ void to_float(float o[8]) const {
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = __uint_as_float(w[i] << 16);
                o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
            const unsigned w2[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i] = __uint_as_float(w[i]); o[4 + i] = __uint_as_float(w2[i]); }
        }
    }
};

// OM: how the attention output is stored - 0 row-major T, 1 the packed A-fragment order of T (bf16 / fp32 builds, batch > 8),
// 2 split planes (the split build: 3 bf16 planes hi + mid + lo in A-fragment order, csrc/decode.hip), 3 two fp16 planes hi + lo
// (MMX_H2, csrc/decode.hip)
// HP: query heads per workgroup.  HP = 2 (the default): the `group` query heads of a kv head are served by ceil(group / 2)
// workgroups, each reading its K / V rows ONCE for two heads - at batch 32 and 7 heads per kv head 256 workgroups (one round of
// the 256 CUs) instead of 448, and 4/7 of the L2 -> CU traffic that bounds this kernel (the cached rows are 205 KB per workgroup
// at 400 fp32 keys: 2.9 us at the ~70 GB/s a CU draws from L2; every head's arithmetic is what it was, in the same order).
template <typename T, int OM, int HP>
__global__ __launch_bounds__(256) void decode_attn_kernel(
    const float* __restrict__ qkv, long ldqkv, int Hq, int Hkv, const float* __restrict__ inv_freq,
    const float* __restrict__ rope_tab, const int32_t* __restrict__ pos, T* __restrict__ kc, T* __restrict__ vc,
    const int32_t* __restrict__ block_table, int max_pages, int page, float scale, T* __restrict__ out, long ldo,
    int nseq) {
    // Single pass ("flash decoding" inside one workgroup): thread = (key group kg of 32, channel chunk dc of 8).
    // For each of its keys a thread loads 16 B of the K row and 16 B of the V row (both in flight together), the 8
    // threads of a key reduce q.k with 3 xor-shuffles, every key group keeps its own running (max, sum, acc[8]) per head;
    // the 32 groups are merged once through LDS.  Two workgroup barriers in total; the block-table row is staged
    // in LDS up front so no load depends on another load except through `pos`.
    constexpr int D = 64, HALF = 32, NG = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);        // [HP][D] rope(q) * scale * log2(e)
    float* kn = qs + HP * D;                           // [D] new key (rounded through T)
    float* vn = kn + D;                                // [D] new value
    float* gm = vn + D;                                // [HP][NG] group max
    float* gl = gm + HP * NG;                          // [HP][NG] group sum
    float* part = gl + HP * NG;                        // [HP][NG][D]
    int* bts = reinterpret_cast<int*>(part + HP * NG * D);  // [max_pages]
    const int tid = threadIdx.x;
    // XCD-aware work mapping: workgroups go to the 8 XCDs round robin by their linear id, and each XCD has its own L2.
    // The workgroups of the query heads that share one KV head read the same cache rows: they are given ids with the same
    // id % 8, so those rows enter ONE L2 once instead of several (at batch 32 and 300 keys the cache reads
    // of a layer are 4.9 MB unique, 34 MB when every head's XCD misses).
    const int group = Hq / Hkv, nsub = (group + HP - 1) / HP;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int kvg = (slot / nsub) * 8 + xcd;           // (sequence, kv head) pair
    if (kvg >= Hkv * nseq) return;                     // uniform: the grid is padded to a multiple of 8 pairs
    const int b = kvg / Hkv, hk = kvg % Hkv, sub = slot % nsub;
    const int h0 = hk * group + sub * HP, nh = min(HP, group - sub * HP);   // this workgroup's heads h0 .. h0 + nh - 1
    const int32_t* bt = block_table + (long)b * max_pages;
    for (int i = tid; i < max_pages; i += 256) bts[i] = bt[i];
    const int p = pos[b];
    const float* src = qkv + (long)b * ldqkv;
    const float sc2 = scale * 1.44269504088896341f;
    if (tid < (HP + 1) * HALF) {
        const int which = tid >> 5, d = tid & 31;      // 0 .. HP-1: q head h0 + which, HP: k head hk
        const float* x = src + (which < HP ? (h0 + min(which, nh - 1)) * D : (Hq + hk) * D);
        float c, s;
        if (rope_tab) {                                // [pos][cos 0..31 | sin 0..31], computed like HF on the host
            c = rope_tab[(long)p * D + d];
            s = rope_tab[(long)p * D + HALF + d];
        } else {
            const float ang = (float)p * inv_freq[d];
            c = cosf(ang);
            s = sinf(ang);
        }
        const float y0 = x[d] * c - x[d + HALF] * s, y1 = x[d + HALF] * c + x[d] * s;
        if (which < HP) { qs[which * D + d] = y0 * sc2; qs[which * D + d + HALF] = y1 * sc2; }   // (a head beyond nh repeats the last one: never stored)
        else { kn[d] = Cvt<T>::to_f(Cvt<T>::from_f(y0)); kn[d + HALF] = Cvt<T>::to_f(Cvt<T>::from_f(y1)); }
    } else if (tid < (HP + 1) * HALF + D) {
        const int d = tid - (HP + 1) * HALF;
        vn[d] = Cvt<T>::to_f(Cvt<T>::from_f(src[(Hq + Hkv + hk) * D + d]));
    }
    __syncthreads();
    if (sub == 0 && tid < D) {                         // one workgroup per kv head appends to the cache
        const long o = (((long)bts[p / page] * Hkv + hk) * page + p % page) * D + tid;
        kc[o] = Cvt<T>::from_f(kn[tid]);
        vc[o] = Cvt<T>::from_f(vn[tid]);
    }
    const int kg = tid >> 3, dc = tid & 7;
    float qv[HP][8];
#pragma unroll
    for (int hp = 0; hp < HP; ++hp)
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[hp][e] = qs[hp * D + dc * 8 + e];
    float m[HP], l[HP], acc[HP][8];
#pragma unroll
    for (int hp = 0; hp < HP; ++hp) {
        m[hp] = -INFINITY;
        l[hp] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[hp][e] = 0.f;
    }
    auto consume = [&](const float kv[8], const float vv[8]) {
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) {
            // (explicit fused multiply-adds: every instantiation rounds alike, whatever the compiler would contract)
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = __builtin_fmaf(qv[hp][e], kv[e], sdot);
            sdot = group8_sum(sdot);                   // the 8 threads of a key (DPP: no LDS round trip on the key chain)
            const float mn = fmaxf(m[hp], sdot);
            const float al = __builtin_amdgcn_exp2f(m[hp] - mn), pj = __builtin_amdgcn_exp2f(sdot - mn);
            l[hp] = __builtin_fmaf(l[hp], al, pj);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hp][e] = __builtin_fmaf(acc[hp][e], al, pj * vv[e]);
            m[hp] = mn;
        }
    };
    // cached keys: U keys per thread in flight at a time (the loop is a chain of memory round trips otherwise:
    // 37 us at 1500 keys with 4 in flight); one pass covers 32*U keys
    constexpr int U = sizeof(T) == 2 ? 12 : 10;       // (fp32 cache: 10 x 64 B per thread in flight, 320 keys per pass)
    for (int j0 = kg; j0 < p; j0 += NG * U) {
        Raw8<T> rk[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * NG;
            if (j < p) {
                const long o = (((long)bts[j / page] * Hkv + hk) * page + j % page) * D + dc * 8;
                rk[u].load(kc + o);
                rv[u].load(vc + o);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u * NG < p) {
                float kv[8], vv[8];
                rk[u].to_float(kv);
                rv[u].to_float(vv);
                consume(kv, vv);
            }
        }
    }
    if (kg == p % NG) {                                // the new token itself (still in LDS)
        float kv[8], vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { kv[e] = kn[dc * 8 + e]; vv[e] = vn[dc * 8 + e]; }
        consume(kv, vv);
    }
#pragma unroll
    for (int hp = 0; hp < HP; ++hp) {
        if (dc == 0) { gm[hp * NG + kg] = m[hp]; gl[hp * NG + kg] = l[hp]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(hp * NG + kg) * D + dc * 8 + e] = acc[hp][e];
    }
    __syncthreads();
    if (tid < nh * D) {
        const int hp = tid >> 6, d = tid & 63, h = h0 + hp;
        const float* gmh = gm + hp * NG;
        const float* glh = gl + hp * NG;
        const float* ph = part + hp * NG * D;
        float M = -INFINITY;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) M = fmaxf(M, gmh[g2]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) {
            const float w = __builtin_amdgcn_exp2f(gmh[g2] - M);     // empty groups: exp2(-inf) = 0
            L = __builtin_fmaf(glh[g2], w, L);
            o = __builtin_fmaf(ph[g2 * D + d], w, o);
        }
        if constexpr (OM == 2) {
            const int nkb = Hq * D / 32, col = h * D + d;
            const long ps = (long)((nseq + 15) / 16) * nkb * 512;
            const long idx = ((((long)(b >> 4) * nkb + (col >> 5)) * 64) + (((col & 31) >> 3) << 4) + (b & 15)) * 8 + (col & 7);
            bf16_t* pl = reinterpret_cast<bf16_t*>(out);
            const float v = o / L;
            const bf16_t hh = f2bf(v);
            const float r1 = v - bf2f(hh);
            const bf16_t mm = f2bf(r1);
            pl[idx] = hh;
            pl[ps + idx] = mm;
            pl[2 * ps + idx] = f2bf(r1 - bf2f(mm));
        } else if constexpr (OM == 3) {
            const int nkb = Hq * D / 32, col = h * D + d;
            const long ps = (long)((nseq + 15) / 16) * nkb * 512;
            const long idx = ((((long)(b >> 4) * nkb + (col >> 5)) * 64) + (((col & 31) >> 3) << 4) + (b & 15)) * 8 + (col & 7);
            unsigned short* pl = reinterpret_cast<unsigned short*>(out);
            const float v = o / L;
            const _Float16 hh = (_Float16)v;
            pl[idx] = __builtin_bit_cast(unsigned short, hh);
            pl[ps + idx] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)hh));
        } else {
            out[OM == 1 ? act_packed_index<T>(b, h * D + d, Hq * D) : (long)b * ldo + h * D + d] = Cvt<T>::from_f(o / L);
        }
    }
}

// GQA-shared decode attention (bf16 build): ONE workgroup per (sequence, kv head) serves all G query heads of the group,
// so every cached K / V row is read once (the per-head kernel above reads it G = 7 times, from L2 after the first, and
// needs 7 x the workgroups: 448 on 256 CUs at batch 32 - two rounds).
//   * S^T = K Q^T on v_mfma_f32_16x16x32_bf16: the A operand is a 16-key page straight from HBM (lane (key, g) holds 8
//     consecutive channels = one 16-byte load, no LDS staging); the B operand is the G roped queries (rows G..15 zero),
//     split into bf16 hi + lo so the product keeps fp32-grade queries; C layout: lane (head, g) holds 4 keys of a page.
//   * page t goes to wave t % 4; a wave holds up to 8 pages (128 keys) of K and V in registers per round, all loads
//     in flight before the first MFMA; online softmax per wave and round (one max / rescale per 128 keys).
//   * P V on the VALU, shared across heads: lane (key sub-row, 8-channel chunk) loads 16 bytes of a V row once and
//     feeds G accumulators; P reaches those lanes through a wave-private LDS patch [key][8 heads].
//   * the 4 waves x 8 key sub-rows are merged once through LDS.
template <bool OPK, int G>
__global__ __launch_bounds__(256) void decode_attn_gqa_kernel(
    const float* __restrict__ qkv, long ldqkv, int Hq, int Hkv, const float* __restrict__ inv_freq,
    const float* __restrict__ rope_tab, const int32_t* __restrict__ pos, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
    const int32_t* __restrict__ block_table, int max_pages, float scale, bf16_t* __restrict__ out, long ldo, int nseq) {
    constexpr int D = 64, HALF = 32, PAGE = 16, NS = 8;             // NS page slots per wave and round
    static_assert(G <= 8, "one 8-float row of P per key");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* qh = reinterpret_cast<bf16_t*>(smem);                    // [16][D] rope(q) * scale * log2(e), high part
    bf16_t* ql = qh + 16 * D;                                        // [16][D] low part
    bf16_t* kn = ql + 16 * D;                                        // [D] new key
    bf16_t* vn = kn + D;                                             // [D] new value
    float* Pw = reinterpret_cast<float*>(vn + D);                    // [4 waves][NS*16 keys][8 heads]
    float* Aw = Pw + 4 * NS * 16 * 8;                                // [4][8] rescale factor of the round
    float* gm = Aw + 32;                                             // [4][8] wave max
    float* gl = gm + 32;                                             // [4][8] wave sum
    float* part = gl + 32;                                           // [4 waves * 8 sub-rows][G][D]
    int* bts = reinterpret_cast<int*>(part + 32 * G * D);            // [max_pages]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / Hkv, hk = blockIdx.x % Hkv;
    const int32_t* bt = block_table + (long)b * max_pages;
    const int p = pos[b];
    const int NT = p / PAGE + 1;                                     // pages that hold keys 0 .. p
    for (int i = tid; i < NT; i += 256) bts[i] = bt[i];
    for (int i = tid; i < 2 * 16 * D / 2; i += 256) reinterpret_cast<unsigned*>(qh)[i] = 0u;       // rows G..15 of qh / ql
    __syncthreads();
    const float* src = qkv + (long)b * ldqkv;
    const float sc2 = scale * 1.44269504088896341f;
    {
        // tid < G*32: query head tid/32, channel pair (d, d+32); the next 32 threads rope the new key
        const int which = tid >> 5, d = tid & 31;
        if (which <= G) {
            const float* x = src + (which < G ? (hk * G + which) * D : (Hq + hk) * D);
            float c, s;
            if (rope_tab) {
                c = rope_tab[(long)p * D + d];
                s = rope_tab[(long)p * D + HALF + d];
            } else {
                const float ang = (float)p * inv_freq[d];
                c = cosf(ang);
                s = sinf(ang);
            }
            const float y0 = x[d] * c - x[d + HALF] * s, y1 = x[d + HALF] * c + x[d] * s;
            if (which < G) {
                const float a0 = y0 * sc2, a1 = y1 * sc2;
                const bf16_t h0 = f2bf(a0), h1 = f2bf(a1);
                qh[which * D + d] = h0;
                qh[which * D + d + HALF] = h1;
                ql[which * D + d] = f2bf(a0 - bf2f(h0));
                ql[which * D + d + HALF] = f2bf(a1 - bf2f(h1));
            } else {
                kn[d] = f2bf(y0);
                kn[d + HALF] = f2bf(y1);
            }
        }
        if (tid < D) vn[tid] = f2bf(src[(Hq + Hkv + hk) * D + tid]);
    }
    __syncthreads();
    if (tid < D) {                                                   // append the new token to the cache
        const long o = (((long)bts[p / PAGE] * Hkv + hk) * PAGE + p % PAGE) * D + tid;
        kc[o] = kn[tid];
        vc[o] = vn[tid];
    }
    short8_t bqh[2], bql[2];                                         // B operand: lane (head l16, g) -> channels ks*32 + 8g ..
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bqh[ks] = *reinterpret_cast<const short8_t*>(qh + l16 * D + ks * 32 + 8 * g);
        bql[ks] = *reinterpret_cast<const short8_t*>(ql + l16 * D + ks * 32 + 8 * g);
    }
    const int ksub = lane >> 3, dc = lane & 7;                       // P V role: key sub-row (0..7), channel chunk
    const int tp = p / PAGE, rp = p % PAGE;                          // where the new token sits
    float* Pme = Pw + wave * NS * 16 * 8;
    float m_run = -INFINITY, l_part = 0.f;                           // lane (head l16, g): running max, partial sum
    typedef __attribute__((ext_vector_type(2))) float float2_t;
    float2_t acc[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[h][e] = float2_t{0.f, 0.f};

    for (int r0 = 0; r0 < NT; r0 += 4 * NS) {
        if (r0 + wave >= NT) break;                                  // wave-uniform: no page left for this wave
        uint4 rk[NS][2], rv[NS][2];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int t = r0 + 4 * i + wave;
            if (t < NT) {
                const long base = ((long)bts[t] * Hkv + hk) * PAGE * D;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) rk[i][ks] = *reinterpret_cast<const uint4*>(kc + base + l16 * D + ks * 32 + 8 * g);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) rv[i][hf] = *reinterpret_cast<const uint4*>(vc + base + (ksub + 8 * hf) * D + dc * 8);
            }
        }
        float4_t sacc[NS];
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int t = r0 + 4 * i + wave;
            sacc[i] = float4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (t < NT) {
                if (t == tp) {                                       // the new token's row comes from LDS (its cache write is not ordered with these loads)
                    if (l16 == rp) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) rk[i][ks] = *reinterpret_cast<const uint4*>(kn + ks * 32 + 8 * g);
                    }
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        if (ksub + 8 * hf == rp) rv[i][hf] = *reinterpret_cast<const uint4*>(vn + dc * 8);
                        // rows beyond the new token have P = 0 but hold whatever the page held: keep NaN / inf out
                        if (ksub + 8 * hf > rp) rv[i][hf] = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
                float4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const short8_t ak = __builtin_bit_cast(short8_t, rk[i][ks]);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bqh[ks], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bql[ks], c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = (t * PAGE + 4 * g + r <= p) ? c[r] : -INFINITY;
                    sacc[i][r] = x;
                    bm = fmaxf(bm, x);
                }
            }
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m_run, bm);                        // finite: the wave's first page has a key <= p
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (r0 + 4 * i + wave < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __builtin_amdgcn_exp2f(sacc[i][r] - m_new);
                    rs += pj;
                    if (l16 < 8) Pme[(i * 16 + 4 * g + r) * 8 + l16] = pj;
                }
            }
        }
        l_part = l_part * alpha + rs;
        if (l16 < 8 && g == 0) Aw[wave * 8 + l16] = alpha;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const float4 a0 = *reinterpret_cast<const float4*>(Aw + wave * 8), a1 = *reinterpret_cast<const float4*>(Aw + wave * 8 + 4);
            const float al[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int h = 0; h < G; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[h][e] *= al[h];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (r0 + 4 * i + wave < NT) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float* pr = Pme + (i * 16 + ksub + 8 * hf) * 8;
                    const float4 p0 = *reinterpret_cast<const float4*>(pr), p1 = *reinterpret_cast<const float4*>(pr + 4);
                    const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                    const unsigned w[4] = {rv[i][hf].x, rv[i][hf].y, rv[i][hf].z, rv[i][hf].w};
                    float2_t vv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) vv[e] = float2_t{__uint_as_float(w[e] << 16), __uint_as_float(w[e] & 0xffff0000u)};
#pragma unroll
                    for (int h = 0; h < G; ++h) {
                        const float2_t ph = {pv[h], pv[h]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[h][e] = __builtin_elementwise_fma(ph, vv[e], acc[h][e]);   // v_pk_fma_f32
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // Pme / Aw are rewritten by the next round
        __builtin_amdgcn_wave_barrier();
    }
    l_part += __shfl_xor(l_part, 16, 64);
    l_part += __shfl_xor(l_part, 32, 64);
    if (l16 < 8 && g == 0) { gm[wave * 8 + l16] = m_run; gl[wave * 8 + l16] = l_part; }
#pragma unroll
    for (int h = 0; h < G; ++h) {
        float* pp = part + ((long)(wave * 8 + ksub) * G + h) * D + dc * 8;
        *reinterpret_cast<float4*>(pp) = make_float4(acc[h][0].x, acc[h][0].y, acc[h][1].x, acc[h][1].y);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[h][2].x, acc[h][2].y, acc[h][3].x, acc[h][3].y);
    }
    __syncthreads();
    for (int idx = tid; idx < G * D; idx += 256) {
        const int h = idx / D, d = idx % D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, gm[w * 8 + h]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float wt = __builtin_amdgcn_exp2f(gm[w * 8 + h] - M);  // a wave without pages: exp2(-inf) = 0
            L += gl[w * 8 + h] * wt;
            float sacc2 = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) sacc2 += part[((long)(w * 8 + k2) * G + h) * D + d];
            o += sacc2 * wt;
        }
        const int col = (hk * G + h) * D + d;
        out[OPK ? act_packed_index<bf16_t>(b, col, Hq * D) : (long)b * ldo + col] = f2bf(o / L);
    }
}
extern "C" int mmx_decode_attn(const float* qkv, int64_t ldqkv, int B, int Hq, int Hkv, int D, const float* inv_freq,
                               const float* rope_tab, const int32_t* pos, void* kc, void* vc, const int32_t* block_table, int max_pages,
                               int page, float scale, void* out, int64_t ldo, int dtype, int out_packed, hipStream_t stream) {
    dtype = MMX_ACT_DTYPE(dtype);
    MMX_CHECK_ARG(qkv && (inv_freq || rope_tab) && pos && kc && vc && block_table && out && B > 0 && D == 64 && Hq % Hkv == 0 && page > 0);
    MMX_CHECK_ARG(out_packed >= 0 && out_packed <= 31);
    const bool gqa_shared = !(out_packed & 2);         // bit 1: force the per-head kernel (A/B measurements, tests)
    const bool split_out = out_packed & 4;             // bit 2: output as split planes (fp32 build of the kernel only)
    const bool split_h2 = out_packed & 8;              // bit 3: output as two fp16 planes (MMX_H2; fp32 build of the kernel only)
    // bit 4: one query head per workgroup whatever the batch (measurements, tests).  Default: two heads per workgroup only where one
    // head per workgroup needs more than one round of the 256 CUs (batch > 18 at 14 heads: the kernel is bound by what a CU draws
    // from L2 there - 882 against 916 us per decode step at batch 32); below that a workgroup's own latency chain is the bound and
    // the second head only lengthens it
    const bool one_head = (out_packed & 16) || (long)B * Hq <= 256;
    MMX_CHECK_ARG(!(split_out || split_h2) || (dtype == MMX_F32 && !(out_packed & 1) && !(split_out && split_h2)));
    out_packed &= 1;
    const size_t max_ctx = (size_t)max_pages * page;
    (void)max_ctx;
    const int hp = one_head ? 1 : 2, nsub = (Hq / Hkv + hp - 1) / hp;
    size_t lds = ((size_t)hp * 64 + 2 * 64 + 2 * hp * 32 + (size_t)hp * 32 * 64 + (size_t)max_pages) * 4;
    MMX_CHECK_ARG(lds <= 64 * 1024);
    dim3 grid(8 * ((Hkv * B + 7) / 8) * nsub);
#define DA(T, OM) do { if (one_head) hipLaunchKernelGGL((decode_attn_kernel<T, OM, 1>), grid, dim3(256), lds, stream, qkv, ldqkv, Hq, Hkv, inv_freq, rope_tab, pos, (T*)kc, (T*)vc, block_table, max_pages, page, scale, (T*)out, ldo, B); \
                       else hipLaunchKernelGGL((decode_attn_kernel<T, OM, 2>), grid, dim3(256), lds, stream, qkv, ldqkv, Hq, Hkv, inv_freq, rope_tab, pos, (T*)kc, (T*)vc, block_table, max_pages, page, scale, (T*)out, ldo, B); } while (0)
    if (dtype == MMX_BF16 && page == 16 && Hq == 7 * Hkv && gqa_shared) {
        const size_t lds2 = (2 * 16 * 64 + 2 * 64) * 2 + (4 * 8 * 16 * 8 + 3 * 32 + 32 * 7 * 64 + (size_t)max_pages) * 4;
        MMX_CHECK_ARG(lds2 <= 160 * 1024);
        if (out_packed) MMX_LDS_OPT_IN((decode_attn_gqa_kernel<true, 7>), lds2);
        else MMX_LDS_OPT_IN((decode_attn_gqa_kernel<false, 7>), lds2);
#define DG(OPK) hipLaunchKernelGGL((decode_attn_gqa_kernel<OPK, 7>), dim3(Hkv * B), dim3(256), lds2, stream, qkv, ldqkv, Hq, Hkv, inv_freq, rope_tab, pos, (bf16_t*)kc, (bf16_t*)vc, block_table, max_pages, scale, (bf16_t*)out, ldo, B)
        if (out_packed) DG(true); else DG(false);
    } else if (dtype == MMX_BF16) { if (out_packed) DA(bf16_t, 1); else DA(bf16_t, 0); }
    else if (dtype == MMX_F32) { if (split_h2) DA(float, 3); else if (split_out) DA(float, 2); else if (out_packed) DA(float, 1); else DA(float, 0); }
    else return MMX_EARG;
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
