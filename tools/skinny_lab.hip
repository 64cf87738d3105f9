// Kernel lab for the LM decode GEMM (csrc/llm.hip: skinny_gemm_kernel), standalone (hipcc, no torch):
// a trimmed bf16 copy of the production kernel with -D switches that remove or reshape one cost at a time, timed as a
// chain of dependent launches on one stream (what the decode graph is).  Findings go back into csrc/llm.hip by hand.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/skinny_lab.hip -o gpurun_out/skinny_lab [-DLAB_...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#ifndef LAB_KS1
#define LAB_KS1 10
#endif
#ifndef LAB_KS2
#define LAB_KS2 7
#endif
#define LAB_KS (MT == 1 ? LAB_KS1 : LAB_KS2)
#ifndef LAB_WAVES
#define LAB_WAVES 4
#endif

__device__ inline bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fff + ((u >> 16) & 1);
    return (bf16_t)(u >> 16);
}

// EPI 1: SwiGLU (NB = 2), EPI 2: residual (NB = 1)
template <int MT, int EPI, int KS>
__global__ __launch_bounds__(512) void lab_kernel(const bf16_t* __restrict__ x, long ldx, int B, int K, int N,
                                                  const bf16_t* __restrict__ wp, float eps, float* __restrict__ outf,
                                                  bf16_t* __restrict__ outa, int ksplit, int ntiles) {
    constexpr int E = 8, KB = 32, NB = EPI == 1 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
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
    const bool rs = EPI == 1;

    float4_t acc[NB][MT];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = float4_t{0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ssq[m] = 0.f;
    float pre_res[MT][4];
    if (active && ksl == 0 && EPI == 2) {
        const int n = tile * 16 + l16;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + 4 * g + r;
                pre_res[m][r] = (row < B && n < N) ? outf[(long)row * N + n] : 0.f;
            }
    }
#ifdef LAB_X_LDS
    // the block's x slice is staged once in LDS (row-major, +16 B row pad) and read back as A fragments
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem + 64 * 1024);
    const int XS_LD = K + 8;
    for (int i = threadIdx.x; i < B * (K / 8); i += blockDim.x) {
        int r = i / (K / 8), c = i % (K / 8);
        *reinterpret_cast<u32x4_t*>(xs + (long)r * XS_LD + c * 8) = *reinterpret_cast<const u32x4_t*>(x + (long)r * ldx + c * 8);
    }
    __syncthreads();
#endif
    if (active) {
        const bf16_t* wbase = wp + ((long)tile * NB * nkb) * 64 * E + (long)lane * E;
        for (int kc = kb0; kc < kb1; kc += KS) {
            u32x4_t wf[KS][NB];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    if (kc + i < kb1)
                        wf[i][n] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wbase + ((long)n * nkb + kc + i) * 64 * E));
            u32x4_t xr[KS][MT];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#if defined(LAB_X_PACKED)
                    // x pre-packed in A-fragment order [m][kb][lane][8]: one contiguous 1 KiB per wave-instruction
                    xr[i][m] = (kc + i < kb1) ? *reinterpret_cast<const u32x4_t*>(x + (((long)m * nkb + kc + i) * 64 + lane) * E)
                                              : u32x4_t{0, 0, 0, 0};
#elif defined(LAB_NO_X)
                    xr[i][m] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)(kc + i + m)};
#elif defined(LAB_X_LDS)
                    xr[i][m] = (kc + i < kb1 && m * 16 + l16 < B)
                                   ? *reinterpret_cast<const u32x4_t*>(xs + (long)(m * 16 + l16) * XS_LD + (kc + i) * KB + g * E)
                                   : u32x4_t{0, 0, 0, 0};
#else
                    xr[i][m] = (kc + i < kb1 && m * 16 + l16 < B)
                                   ? *reinterpret_cast<const u32x4_t*>(x + (long)(m * 16 + l16) * ldx + (kc + i) * KB + g * E)
                                   : u32x4_t{0, 0, 0, 0};
#endif
                }
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                if (kc + i >= kb1) continue;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    u32x4_t raw = xr[i][m];
#ifndef LAB_NO_SSQ
                    if (rs) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            float xv = __uint_as_float((raw[e / 2] >> ((e & 1) * 16)) << 16);
                            ssq[m] += xv * xv;
                        }
                    }
#endif
                    short8_t af = *reinterpret_cast<const short8_t*>(&raw);
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        short8_t bfr = *reinterpret_cast<const short8_t*>(&wf[i][n]);
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[n][m], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (rs) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ssq[m] += __shfl_xor(ssq[m], 16, 64);
            ssq[m] += __shfl_xor(ssq[m], 32, 64);
        }
    }
    constexpr int PER = NB * MT * 4 + MT;
#ifndef LAB_NO_REDUCE
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
#else
    if (ksl != 0) return;
#endif
    if (!active) return;
#ifdef LAB_NO_EPI
    if (acc[0][0][0] == 123.456f) outa[0] = 1;
    return;
#endif
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
            const int n = tile * 16 + l16;
            if (n >= N) continue;
            if constexpr (EPI == 1) {
                float gte = acc[0][m][r] * sc, up = acc[1][m][r] * sc;
                float sl = gte / (1.f + __expf(-gte));
                outa[(long)row * N + n] = f2bf(sl * up);
            } else {
                float v = acc[0][m][r] * sc + pre_res[m][r];
                outf[(long)row * N + n] = v;
                outa[(long)row * N + n] = f2bf(v);
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MT, int EPI>
static float run_case(int B, int K, int N, int layers, int reps, const char* tag) {
    constexpr int NB = EPI == 1 ? 2 : 1;
    const int ntiles = (N + 15) / 16, nkb = K / 32;
    const size_t wbytes = (size_t)ntiles * NB * nkb * 64 * 8 * 2;
    std::vector<bf16_t*> ws(layers);
    std::vector<bf16_t> h(wbytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 20) % 512);   // small positive bf16
    for (int l = 0; l < layers; ++l) { CK(hipMalloc(&ws[l], wbytes)); CK(hipMemcpy(ws[l], h.data(), wbytes, hipMemcpyHostToDevice)); }
    bf16_t *x, *outa; float* outf;
    CK(hipMalloc(&x, (size_t)64 * K * 2)); CK(hipMalloc(&outa, (size_t)64 * N * 2)); CK(hipMalloc(&outf, (size_t)64 * N * 4));
    CK(hipMemset(x, 0x3c, (size_t)64 * K * 2)); CK(hipMemset(outf, 0, (size_t)64 * N * 4));
    int ksplit = 1;
    while (ksplit < 8 && (nkb + ksplit - 1) / ksplit > LAB_KS) ksplit *= 2;
    int waves = ksplit >= LAB_WAVES ? ksplit : LAB_WAVES;
    int tpb = waves / ksplit;
    dim3 grid((ntiles + tpb - 1) / tpb), block(waves * 64);
    size_t lds = (size_t)waves * (NB * MT * 4 + MT) * 64 * 4;
#ifdef LAB_X_LDS
    lds = 64 * 1024 + (size_t)B * (K + 8) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_kernel<MT, EPI, LAB_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int l) {
        hipLaunchKernelGGL((lab_kernel<MT, EPI, LAB_KS>), grid, block, lds, s, x, (long)K, B, K, N, ws[l], 1e-6f, outf, outa, ksplit, ntiles);
    };
    for (int l = 0; l < layers; ++l) launch(l);
    CK(hipStreamSynchronize(s));
    // capture the chain in a graph, as the decode step is
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < reps; ++r) for (int l = 0; l < layers; ++l) launch(l);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    float us = ms * 1e3f / (reps * layers);
    printf("%-28s B=%2d K=%4d N=%4d epi=%d grid=%4d x %3d thr ksplit=%d: %6.2f us  %7.1f GB/s\n", tag, B, K, N, EPI, grid.x, block.x, ksplit,
           us, wbytes / us / 1e3);
    for (int l = 0; l < layers; ++l) CK(hipFree(ws[l]));
    CK(hipFree(x)); CK(hipFree(outa)); CK(hipFree(outf));
    return us;
}

int main(int argc, char** argv) {
    const char* tag = argc > 1 ? argv[1] : "lab";
    run_case<1, 1>(1, 896, 4864, 24, 10, tag);
    run_case<2, 1>(32, 896, 4864, 24, 10, tag);
    run_case<1, 2>(1, 4864, 896, 24, 10, tag);
    run_case<2, 2>(32, 4864, 896, 24, 10, tag);
    return 0;
}
