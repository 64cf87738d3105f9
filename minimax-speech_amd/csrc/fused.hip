// Row-tile fused kernels of the CFM estimator (speech/cosyvoice/flow/decoder.py:405-496 with
// matcha/models/components/transformer.py:243-316 and decoder.py:56-61 of the reference) for gfx950.
//
// The estimator is 14 x (causal ResNet block + 4 transformer blocks) at full time resolution with C = 256 channels:
// per block a chain of short-K GEMMs (K = 256 .. 1536) over M = B*T rows.  As separate launches every link of the
// chain is latency bound (5-15 us per launch at M = 1 000 .. 14 000, 8 launches per transformer block, MFMA busy
// 10 %: profiles/r01_pmc_flow_kernels.txt).  Only attention mixes rows; everything between two attentions is
// row-local.  So one workgroup takes a tile of BM rows through the whole row-local chain:
//
//   est_tail_kernel   : attn-out projection + residual -> LayerNorm -> FF1 + GELU -> FF2 + residual
//                       -> [LayerNorm of the NEXT block -> its Q/K/V projection]  or  [masked activation copy]
//   est_resnet_kernel : causal conv k3 -> LayerNorm -> Mish (+ time embedding) -> causal conv k3 -> LayerNorm -> Mish
//                       -> + 1x1 residual conv -> LayerNorm of the next block -> its Q/K/V projection
//
// A transformer block is then 2 launches (this + flash attention) instead of 8, a ResNet block 1 instead of 5, and
// the fp32 residual stream, the LayerNorm outputs and the 1024-wide FF intermediate never leave the CU.
//
// Data flow inside a workgroup (256 threads = 4 waves, one per SIMD):
//   * the A operand of every GEMM stage is a row tile resident in LDS (full K, row pitch = 32 B mod 256 B so that
//     the ds_read_b128 fragment reads are bank-conflict free: MI355X_MICROARCH.md, LDS);
//   * weights are pre-packed in MFMA B-fragment order ([n/16][k/KB][64 lanes][16 B], mmx_pack_skinny) and streamed
//     from L2 straight into registers, 1 KiB per wave-instruction, through a 2-deep register ring that runs ahead
//     across stage boundaries (the next stage's first fragments are in flight during the current epilogue);
//   * every wave owns a 64-column slice of each stage's N; accumulators (MFMA C layout: lane = column) are turned
//     into the row layout (lane = 16 consecutive columns of one row) through a private LDS patch, so LayerNorm
//     reductions are in-lane + 2 shuffles + one LDS exchange across the 4 waves, and every global / LDS store is a
//     16-byte access;
//   * FF1 -> FF2 is chunked over the intermediate (2 x 512 columns): FF2 accumulates chunk by chunk, the
//     intermediate lives in the LDS buffer the attention output occupied.
//
// bf16 build: v_mfma_f32_16x16x32_bf16; fp32 parity build: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), fp32 tiles.
// Split build (NS = 2, MMX_X2): bf16 weights as in the bf16 build; every activation that is a GEMM operand lives in LDS as
// TWO bf16 planes (hi = bf16(x), lo = bf16(x - hi)) and costs two MFMAs per weight fragment; activations in HBM (attention
// output in, Q | K | V out, the masked copy) are fp32.  The weight stream from L2 - what bounds these kernels - is unchanged.
#include "common.h"
#include "../../include/mmx_hip.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// lab build only (common.h, MMX_LAB; set by mmx_lab_tail_stamps): est_tail_kernel writes shader-clock stamps
// [workgroup][wave][64] at its stage boundaries (tools/tail_lab.py --stamps).  The product build compiles no stamp.
__device__ unsigned long long* g_tail_stamps = nullptr;
#define TSTAMP(i) do { if constexpr (LAB) { if (st && lane == 0) st[(i)] = __builtin_amdgcn_s_memtime(); } } while (0)

template <typename T> struct FT;
template <> struct FT<bf16_t> { static constexpr int E = 8, KB = 32; typedef short8_t frag_t; };
template <> struct FT<float> { static constexpr int E = 4, KB = 16; typedef float4_t frag_t; };

template <typename T>
__device__ __forceinline__ float4_t mma(typename FT<T>::frag_t a, typename FT<T>::frag_t b, float4_t c) {
    if constexpr (sizeof(T) == 2) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    }
}

// LDS row pitch in bytes of a [rows][K] tile of T: 32 B mod 256 B (conflict-free fragment reads)
__host__ __device__ constexpr int tile_pitch(int K, int esz) { return ((K * esz + 255) & ~255) + 32; }

// ---------------------------------------------------------------------------------------------------------------
// One GEMM stage of a wave: acc[MF][NF] += A(LDS row tile) x W(packed fragments, NF n-fragments of 16 columns).
// The weight ring holds PF k-steps ahead; run() refills it from THIS stage while k-steps remain and from the NEXT
// stage afterwards, so a stage starts with its first fragments already in registers.
// PF: k-steps the ring runs ahead.  One k-step of a wave is MF*NF MFMAs (16 cycles each), an L2 hit takes ~500-800
// cycles: a 16-row tile (MF = 1) needs ~8 k-steps in flight, a 64-row tile 2.  Every stage's k-step count must be a
// multiple of PF (checked on the host).
// The ring is NF (4, or 2 in the narrow-pass 8-wave kernels) n-fragments wide; a stage that owns fewer (nf = 2: 32 columns
// per wave in the 8-wave kernels) uses the first nf slots, and head loads for the unused slots repeat fragment 0 (no branch
// around a load).
template <typename T, int NF, int PF_>
struct WRing {
    static constexpr int E = FT<T>::E, PF = PF_;
    u32x4_t w[PF][NF];
    __device__ __forceinline__ void prime(const T* wb, long ns, int nk, int nf) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                w[p][j] = *reinterpret_cast<const u32x4_t*>(wb + (j < nf ? j : 0) * ns + (long)(p < nk ? p : nk - 1) * 64 * E);
    }
};

// a_lane: LDS byte address of this lane's 16-byte chunk of (m-fragment 0, row l16, k-step 0): tile + l16*pitch + g*16.
// K is walked as `taps` groups of `cin_steps` k-steps; tap t reads the tile `t` rows further down (causal conv k3:
// taps = 3; Linear: taps = 1).  wb / ns / nk: packed weights of this stage for this wave (lane offset included),
// elements between n-fragments, k-steps (even).  wbn / nsn: the next stage's (NULL: none).
// NS / plane: the A tile is NS planes, `plane` bytes apart (split build); every plane is multiplied with the same B fragment.
template <typename T, int MF, int NF, int PF, int NS = 1, int RW = 4>
__device__ __forceinline__ void stage_run(WRing<T, RW, PF>& ring, const char* a_lane, int pitch, int cin_steps,
                                          const T* wb, long ns, int nk, const T* wbn, long nsn, int nkn, int nfn,
                                          float4_t (&acc)[MF][NF], int plane = 0, int tap_pitch = 0) {
    constexpr int E = FT<T>::E, KB = FT<T>::KB;
    typedef typename FT<T>::frag_t frag_t;
    if (tap_pitch == 0) tap_pitch = pitch;             // dilated convs (DAC): tap t reads t * dilation rows further down
    // A fragments come from LDS AD - 1 k-steps ahead of their use (a k-step is MF*NF MFMAs = 64 cycles at MF = 1,
    // 256 at MF = 4; the LDS latency is ~130 cycles)
    constexpr int AD = (PF % 2) ? PF : ((MF == 1 && PF >= 4) ? 4 : 2);
    static_assert(PF % AD == 0, "ring depths");
    // 64-row split tile (two planes x four m-fragments): the A fragments are read per (k-step, plane) unit, one unit ahead of
    // its MFMAs - 32 instead of 64 registers (the k-step-ahead form spilled 76 bytes per lane).  Same MFMA order per accumulator.
    if constexpr (NS == 2 && MF == 4) {
        frag_t a2[2][MF];
        int rt = 0, rc = 0, rs = 0, rp = 0;
        auto read_u = [&](frag_t (&dst)[MF]) {
            const char* ap = a_lane + rt * tap_pitch + rc * (KB * (int)sizeof(T)) + rp * plane;
#pragma unroll
            for (int i = 0; i < MF; ++i) dst[i] = *reinterpret_cast<const frag_t*>(ap + i * 16 * pitch);
            if (rp == 0) rp = 1;
            else {
                rp = 0;
                if (rs + 1 < nk) {                         // past the end: re-read the last k-step (never used)
                    ++rs;
                    if (++rc == cin_steps) { rc = 0; ++rt; }
                }
            }
        };
        auto group2 = [&](int ks, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                frag_t b[NF];
#pragma unroll
                for (int jj = 0; jj < NF; ++jj) b[jj] = __builtin_bit_cast(frag_t, ring.w[p][jj]);
                if constexpr (LAST) {
                    if (wbn) {
                        const T* src = wbn + (long)(p < nkn ? p : nkn - 1) * 64 * E;
#pragma unroll
                        for (int jj = 0; jj < RW; ++jj) ring.w[p][jj] = *reinterpret_cast<const u32x4_t*>(src + (jj < nfn ? jj : 0) * nsn);
                    }
                } else {
                    const T* src = wb + (long)(ks + p + PF) * 64 * E;
#pragma unroll
                    for (int jj = 0; jj < NF; ++jj) ring.w[p][jj] = *reinterpret_cast<const u32x4_t*>(src + jj * ns);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    read_u(a2[(s2 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int jj = 0; jj < NF; ++jj) acc[i][jj] = mma<T>(a2[s2][i], b[jj], acc[i][jj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        read_u(a2[0]);
        for (int ks = 0; ks + PF < nk; ks += PF) group2(ks, std::false_type{});
        group2(nk - PF, std::true_type{});
        return;
    }
    frag_t a[AD][NS][MF];
    int rt = 0, rc = 0, rs = 0;                        // tap / k-step inside the tap / k-step of the next A read
    auto read_a = [&](frag_t (&dst)[NS][MF]) {
        const char* ap = a_lane + rt * tap_pitch + rc * (KB * (int)sizeof(T));
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
            for (int i = 0; i < MF; ++i) dst[s2][i] = *reinterpret_cast<const frag_t*>(ap + s2 * plane + i * 16 * pitch);
        if (rs + 1 < nk) {                             // past the end: re-read the last k-step (never used)
            ++rs;
            if (++rc == cin_steps) { rc = 0; ++rt; }
        }
    };
    // one group of PF k-steps.  LAST: the refills of this group are the head of the NEXT stage (k-steps 0 .. PF-1 of wbn,
    // nfn fragments wide); otherwise k-steps ks+PF .. of this stage, NF fragments wide.  No load sits under a branch
    // that depends on the k-step, so the compiler keeps counted vmcnt waits.
    auto group = [&](int ks, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            read_a(a[(p + AD - 1) % AD]);
            frag_t b[NF];
#pragma unroll
            for (int jj = 0; jj < NF; ++jj) b[jj] = __builtin_bit_cast(frag_t, ring.w[p][jj]);
            if constexpr (LAST) {
                if (wbn) {                             // uniform per stage
                    const T* src = wbn + (long)(p < nkn ? p : nkn - 1) * 64 * E;
#pragma unroll
                    for (int jj = 0; jj < RW; ++jj) ring.w[p][jj] = *reinterpret_cast<const u32x4_t*>(src + (jj < nfn ? jj : 0) * nsn);
                }
            } else {
                const T* src = wb + (long)(ks + p + PF) * 64 * E;
#pragma unroll
                for (int jj = 0; jj < NF; ++jj) ring.w[p][jj] = *reinterpret_cast<const u32x4_t*>(src + jj * ns);
            }
            // keep the refill (and the A read-ahead) HERE: left alone, the scheduler sinks these loads down to their
            // first use PF k-steps later (register pressure), which turns the ring into load-then-wait every k-step
            // (measured: the same kernel time at ring depth 2, 4 and 8)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int jj = 0; jj < NF; ++jj) acc[i][jj] = mma<T>(a[p % AD][s2][i], b[jj], acc[i][jj]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int s_ = 0; s_ < AD - 1; ++s_) read_a(a[s_]);
    for (int ks = 0; ks + PF < nk; ks += PF) group(ks, std::false_type{});
    group(nk - PF, std::true_type{});
}

// Weight planes (MMX_X2W: the weights of an fp32 checkpoint as bf16 hi + lo, the two packs one after the other, `wlo` elements
// apart): a stage runs TWICE - the hi pack against both activation planes, then the lo pack against the activation's hi plane
// (NS = 1 over plane 0) - into the same accumulator: hi*hi + lo_a*hi_w + hi_a*lo_w, three MFMAs per fragment pair; the term
// dropped (lo*lo) is below the last kept bit of either operand.  The ring chains hi -> lo -> the next stage's hi.
template <typename T, int MF, int NF, int PF, int NS, int RW, bool WP>
__device__ __forceinline__ void stage_run_w(WRing<T, RW, PF>& ring, const char* a_lane, int pitch, int cin_steps,
                                            const T* wb, long ns, int nk, long wlo, const T* wbn, long nsn, int nkn, int nfn,
                                            float4_t (&acc)[MF][NF], int plane = 0, int tap_pitch = 0) {
    if constexpr (!WP) {
        stage_run<T, MF, NF, PF, NS, RW>(ring, a_lane, pitch, cin_steps, wb, ns, nk, wbn, nsn, nkn, nfn, acc, plane, tap_pitch);
    } else {
        stage_run<T, MF, NF, PF, NS, RW>(ring, a_lane, pitch, cin_steps, wb, ns, nk, wb + wlo, ns, nk, NF, acc, plane, tap_pitch);
        stage_run<T, MF, NF, PF, 1, RW>(ring, a_lane, pitch, cin_steps, wb + wlo, ns, nk, wbn, nsn, nkn, nfn, acc, 0, tap_pitch);
    }
}

template <int MF, int NF>
__device__ __forceinline__ void zero_acc(float4_t (&acc)[MF][NF]) {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
}

// MFMA C layout (lane = column l16 of n-fragment j, rows 4g..4g+3) -> row layout: lane owns row (lane >> 2) of the
// 16-row fragment and NF*4 consecutive columns (lane & 3)*NF*4 .. of the wave's NF*16-column slice, through the wave's
// private fp32 patch (16 rows of NF*16 + 4 floats: conflict-free ds_write_b32).
constexpr int PATCH_FLOATS = 16 * 68;                  // per wave: the widest slice is 64 columns
constexpr int TAIL_PRM_FLOATS = 6 * 256 + 1024;        // est_tail_kernel's bias / LayerNorm vectors in LDS
template <int NF>
__device__ __forceinline__ void to_rows(const float4_t (&acc)[NF], float* patch, int lane, float (&v)[NF * 4]) {
    constexpr int LDC = NF * 16 + 4;
    const int g = lane >> 4, l16 = lane & 15;
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(4 * g + r) * LDC + j * 16 + l16] = acc[j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* src = patch + (lane >> 2) * LDC + (lane & 3) * (NF * 4);
#pragma unroll
    for (int c = 0; c < NF * 4; c += 4) {
        const float4_t t = *reinterpret_cast<const float4_t*>(src + c);
        v[c] = t[0]; v[c + 1] = t[1]; v[c + 2] = t[2]; v[c + 3] = t[3];
    }
    __builtin_amdgcn_wave_barrier();                   // the patch is rewritten by the next fragment
}

// The same conversion through a ONE-fragment patch (16 rows of 16 + 4 floats) for the 64-row split tile, whose two-plane tiles leave
// 10 KB of LDS for all eight patches: fragment j serves the lanes whose 8 columns lie in it ((lane & 3) >> 1 == j).  Same values in
// the same lanes as to_rows<2>.
constexpr int SPATCH_FLOATS = 16 * 20;
__device__ __forceinline__ void to_rows_sp(const float4_t (&acc)[2], float* patch, int lane, float (&v)[8]) {
    constexpr int LDC = 20;
    const int g = lane >> 4, l16 = lane & 15;
    const float* src = patch + (lane >> 2) * LDC + (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(4 * g + r) * LDC + l16] = acc[j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (((lane >> 1) & 1) == j) {
            const float4_t t0 = *reinterpret_cast<const float4_t*>(src), t1 = *reinterpret_cast<const float4_t*>(src + 4);
            v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];
            v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
        }
        __builtin_amdgcn_wave_barrier();               // the patch is rewritten by the next fragment
    }
}
template <int NF, bool SP>
__device__ __forceinline__ void rows_of(const float4_t (&acc)[NF], float* patch, int lane, float (&v)[NF * 4]) {
    if constexpr (SP) { static_assert(NF == 2, "one-fragment patch: 32-column slices"); to_rows_sp(acc, patch, lane, v); }
    else to_rows<NF>(acc, patch, lane, v);
}

template <int N>
__device__ __forceinline__ void loadn(const float* p, float (&v)[N]) {
#pragma unroll
    for (int c = 0; c < N; c += 4) {
        const float4_t t = *reinterpret_cast<const float4_t*>(p + c);
        v[c] = t[0]; v[c + 1] = t[1]; v[c + 2] = t[2]; v[c + 3] = t[3];
    }
}
template <int N>
__device__ __forceinline__ void storen(float* p, const float (&v)[N]) {
#pragma unroll
    for (int c = 0; c < N; c += 4) *reinterpret_cast<float4_t*>(p + c) = float4_t{v[c], v[c + 1], v[c + 2], v[c + 3]};
}
// N (8 or 16) consecutive values as T (global or LDS destination, 16-byte aligned)
template <typename T, int N>
__device__ __forceinline__ void storen_T(T* p, const float (&v)[N]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int c = 0; c < N; c += 8) {
            uint4 pk;
            pk.x = pack_bf16x2(v[c], v[c + 1]);
            pk.y = pack_bf16x2(v[c + 2], v[c + 3]);
            pk.z = pack_bf16x2(v[c + 4], v[c + 5]);
            pk.w = pack_bf16x2(v[c + 6], v[c + 7]);
            *reinterpret_cast<uint4*>(p + c) = pk;
        }
    } else {
        storen<N>(p, v);
    }
}

// N consecutive values into an A tile in LDS at byte address `p`: as T (one plane) or as bf16 hi + lo planes `plane` bytes apart
template <typename T, int NS, int N>
__device__ __forceinline__ void store_tile(char* p, int plane, const float (&v)[N]) {
    if constexpr (NS == 1) {
        storen_T<T, N>(reinterpret_cast<T*>(p), v);
    } else {
        float lo[N];
#pragma unroll
        for (int c = 0; c < N; c += 8) {
            uint4 pk;
            pk.x = pack_bf16x2(v[c], v[c + 1]);
            pk.y = pack_bf16x2(v[c + 2], v[c + 3]);
            pk.z = pack_bf16x2(v[c + 4], v[c + 5]);
            pk.w = pack_bf16x2(v[c + 6], v[c + 7]);
            *reinterpret_cast<uint4*>(p + c * 2) = pk;
            const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[c + 2 * e] = v[c + 2 * e] - __uint_as_float(w[e] << 16);
                lo[c + 2 * e + 1] = v[c + 2 * e + 1] - __uint_as_float(w[e] & 0xffff0000u);
            }
        }
        storen_T<bf16_t, N>(reinterpret_cast<bf16_t*>(p + plane), lo);
    }
}

// LayerNorm of the rows of a [rows x 256] tile held in the row layout (v[i][CW] per wave, NW waves x 256/NW columns).
// Two-pass statistics like torch (mean, then the mean of squared deviations), partial sums exchanged through
// stats[rows][NW].  Contains 3 workgroup barriers; all waves must call it.  Leaves normalised*gamma+beta in v.
// gamma / beta: this lane's CW columns, loaded by the caller BEFORE the preceding MFMA stage (see "epilogue operands").
template <int MFR, int CW, int NW>
__device__ __forceinline__ void layernorm_rows(float (&v)[MFR][CW], float* stats, const float (&g)[CW], const float (&be)[CW],
                                               float eps, int wave, int lane) {
    const int rl = lane >> 2;
    float mean[MFR];
    auto total = [&](int row) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4) {
            const float4_t q = *reinterpret_cast<const float4_t*>(stats + row * NW + w);
            t += (q[0] + q[1]) + (q[2] + q[3]);
        }
        return t;
    };
    // (no barrier in front: every caller has a workgroup barrier between the last read of `stats` by the previous call and this one -
    // the FF chunk barriers in est_tail, the tile barriers in est_resnet)
#pragma unroll
    for (int i = 0; i < MFR; ++i) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) s += v[i][c];
        s += dpp_f32<0xB1>(s);                         // the 4 lanes of a row (DPP quad sums, no LDS round trip)
        s += dpp_f32<0x4E>(s);
        if ((lane & 3) == 0) stats[(i * 16 + rl) * NW + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MFR; ++i) mean[i] = total(i * 16 + rl) * (1.0f / 256.0f);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MFR; ++i) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CW; ++c) { const float d = v[i][c] - mean[i]; q += d * d; }
        q += dpp_f32<0xB1>(q);
        q += dpp_f32<0x4E>(q);
        if ((lane & 3) == 0) stats[(i * 16 + rl) * NW + wave] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MFR; ++i) {
        const float rstd = rsqrtf(total(i * 16 + rl) * (1.0f / 256.0f) + eps);
#pragma unroll
        for (int c = 0; c < CW; ++c) v[i][c] = (v[i][c] - mean[i]) * rstd * g[c] + be[c];
    }
}

// cooperative copy of `rows` rows x K elements (global row stride ld, first row index r0 of `nvalid` valid rows
// starting at `src`; rows outside [0, nvalid) read zero) into an LDS tile with `pitch` bytes per row
// Both copies issue their global loads in batches of U (addresses of rows outside the valid range are clamped and the
// value is replaced by zero afterwards): one load -> wait -> LDS store round trip per 16-byte chunk made the 64 KB
// attention tile of a 64-row workgroup a 6 us prologue (tools/tail_lab.py --stamps).
template <typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ src, long ld, int r0, int nvalid, int rows, int K,
                                          char* tile, int pitch, int tid, int nthreads) {
    constexpr int E = FT<T>::E, U = 8;
    const int cpr = K / E;                             // 16-byte chunks per row
    const int total = rows * cpr;
    for (int base = tid; base < total; base += U * nthreads) {
        uint4 v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int id = base + u * nthreads;
            const int idc = id < total ? id : total - 1;
            const int r = idc / cpr, ch = idc - r * cpr;
            const int gr = r0 + r;
            const bool ok = gr >= 0 && gr < nvalid;
            const int grc = gr < 0 ? 0 : (gr < nvalid ? gr : nvalid - 1);
            v[u] = *reinterpret_cast<const uint4*>(src + (long)grc * ld + ch * E);
            if (!ok) v[u] = make_uint4(0, 0, 0, 0);
            off[u] = id < total ? r * pitch + ch * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (off[u] >= 0) *reinterpret_cast<uint4*>(tile + off[u]) = v[u];
    }
}
// split build: fp32 rows -> bf16 hi + lo planes (`plane` bytes apart), 4 values per 16-byte global chunk
__device__ __forceinline__ void load_tile_split(const float* __restrict__ src, long ld, int r0, int nvalid, int rows, int K,
                                                char* tile, int pitch, int plane, int tid, int nthreads) {
    constexpr int U = 8;
    const int cpr = K / 4;
    const int total = rows * cpr;
    for (int base = tid; base < total; base += U * nthreads) {
        float4 v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int id = base + u * nthreads;
            const int idc = id < total ? id : total - 1;
            const int r = idc / cpr, ch = idc - r * cpr;
            const int gr = r0 + r;
            const bool ok = gr >= 0 && gr < nvalid;
            const int grc = gr < 0 ? 0 : (gr < nvalid ? gr : nvalid - 1);
            v[u] = *reinterpret_cast<const float4*>(src + (long)grc * ld + ch * 4);
            if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            off[u] = id < total ? r * pitch + ch * 8 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (off[u] < 0) continue;
            uint2 hi, lo;
            hi.x = pack_bf16x2(v[u].x, v[u].y);
            hi.y = pack_bf16x2(v[u].z, v[u].w);
            lo.x = pack_bf16x2(v[u].x - __uint_as_float(hi.x << 16), v[u].y - __uint_as_float(hi.x & 0xffff0000u));
            lo.y = pack_bf16x2(v[u].z - __uint_as_float(hi.y << 16), v[u].w - __uint_as_float(hi.y & 0xffff0000u));
            *reinterpret_cast<uint2*>(tile + off[u]) = hi;
            *reinterpret_cast<uint2*>(tile + plane + off[u]) = lo;
        }
    }
}

// the 64-row split tile's attention rows: 512 columns as two K halves into two tiles (columns 0 .. 255 -> tile0, 256 .. 511 -> tile1,
// same pitch / plane distance), every global load of the workgroup's share in flight before the first LDS store (one memory
// round trip for the 128 KB; as two load_tile_split calls it was two)
__device__ __forceinline__ void load_tile_split_halves(const float* __restrict__ src, long ld, int r0, int nvalid, int rows,
                                                       char* tile0, char* tile1, int pitch, int plane, int tid, int nthreads) {
    constexpr int U = 16;
    constexpr int cpr = 512 / 4;
    const int total = rows * cpr;
    for (int base = tid; base < total; base += U * nthreads) {
        float4 v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int id = base + u * nthreads;
            const int idc = id < total ? id : total - 1;
            const int r = idc / cpr, ch = idc - r * cpr;
            const int gr = r0 + r;
            const bool ok = gr >= 0 && gr < nvalid;
            const int grc = gr < 0 ? 0 : (gr < nvalid ? gr : nvalid - 1);
            v[u] = *reinterpret_cast<const float4*>(src + (long)grc * ld + ch * 4);
            if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            off[u] = id < total ? r * pitch + (ch & 63) * 8 + ((ch >> 6) ? (int)(tile1 - tile0) : 0) : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (off[u] < 0) continue;
            uint2 hi, lo;
            hi.x = pack_bf16x2(v[u].x, v[u].y);
            hi.y = pack_bf16x2(v[u].z, v[u].w);
            lo.x = pack_bf16x2(v[u].x - __uint_as_float(hi.x << 16), v[u].y - __uint_as_float(hi.x & 0xffff0000u));
            lo.y = pack_bf16x2(v[u].z - __uint_as_float(hi.y << 16), v[u].w - __uint_as_float(hi.y & 0xffff0000u));
            *reinterpret_cast<uint2*>(tile0 + off[u]) = hi;
            *reinterpret_cast<uint2*>(tile0 + plane + off[u]) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Shared tail of both kernels: x (row layout) -> LayerNorm(n1) -> A1 tile -> Q/K/V projection.
// bf16: Q,K row-major [B][T][1024] and V TRANSPOSED vt[b][512][Tp] (what the flash kernel reads);
// fp32: q|k|v row-major [B][T][1536] (the dense attention kernel reads strided heads).
// Every wave takes PC = 16 * PW-column passes (PW = 4: 64 columns; PW = 2, 8 waves: 32 columns - half the accumulator
// registers, see est_tail_kernel): pass p of wave w is kind p / PPK (Q, K, V), columns kind*512 + (w*PPK + p % PPK)*PC of the
// packed [1536][256] projection, PPK = 512 / (PC * NW) passes per kind.
template <typename T, int NW, int PW = 4>
__device__ __forceinline__ const T* qkv_pass(const void* wqkv, int wave, int lane, int p) {
    constexpr int E = FT<T>::E, KB = FT<T>::KB, PC = 16 * PW, PPK = 512 / (PC * NW);
    return reinterpret_cast<const T*>(wqkv) + (long)lane * E +
           (long)(((p / PPK) * 512 + (wave * PPK + p % PPK) * PC) / 16) * ((long)(256 / KB) * 64 * E);
}

// the weight ring must already hold the head of pass 0 (the caller's last stage chains into qkv_pass(.., 0))
template <typename T, int MF, int PF, int NW, int NS = 1, int PW = 4, bool WP = false>
__device__ __forceinline__ void ln_qkv(float (&xv)[MF][64 / NW], const float (&n1g)[64 / NW], const float (&n1b)[64 / NW],
                                       const MmxEstNext& nx, float eps, char* a1, float* patch, float* stats,
                                       WRing<T, PW, PF>& ring, int b, int t0, int Tn, int wave, int lane, int plane = 0,
                                       unsigned long long* st = nullptr) {
    typedef std::conditional_t<NS == 1, T, float> TI;   // activation type in HBM
    constexpr int E = FT<T>::E, KB = FT<T>::KB, C = 256, CW = 64 / NW, PC = 16 * PW, PPK = 512 / (PC * NW), NP = 3 * PPK;
    constexpr int P1 = tile_pitch(C, sizeof(T));
    constexpr int NK = C / KB;
    const int g = lane >> 4, l16 = lane & 15, rl = lane >> 2;
    const int col0 = wave * (256 / NW) + (lane & 3) * CW;
    const long ns = (long)NK * 64 * E;
    layernorm_rows<MF, CW, NW>(xv, stats, n1g, n1b, eps, wave, lane);
#pragma unroll
    for (int i = 0; i < MF; ++i) store_tile<T, NS, CW>(a1 + (i * 16 + rl) * P1 + col0 * (int)sizeof(T), plane, xv[i]);
    __syncthreads();
    TSTAMP(32);
    const char* a_lane = a1 + l16 * P1 + g * 16;
    // bf16 build: Q | K as bf16 rows and V transposed.  Split build: the same two layouts as bf16 PLANES when the caller
    // gives vt_out (q_out bf16 [B][T][ldq >= 2048] = [hi Q | hi K | lo Q | lo K], vt_out [B][2][512][ldvt]: what
    // mmx_attn_flash_xs reads - the operands are split here, once, instead of in every query tile of the attention), or
    // fp32 rows Q | K | V when vt_out is NULL.
    constexpr bool VTC = sizeof(T) == 2;
    const bool planes = NS > 1 && nx.vt_out != nullptr;
    const bool vt_path = VTC && (NS == 1 || planes);
    constexpr int PV = 16 * 2 + 16;                    // V^T patch pitch: [64 columns][16 frames] of bf16 inside the wave's patch
    for (int p = 0; p < NP; ++p) {
        float4_t acc[MF][PW];
        zero_acc(acc);
        const T* wn = p + 1 < NP ? qkv_pass<T, NW, PW>(nx.wqkv, wave, lane, p + 1) : nullptr;
        stage_run_w<T, MF, PW, PF, NS, PW, WP>(ring, a_lane, P1, NK, qkv_pass<T, NW, PW>(nx.wqkv, wave, lane, p), ns, NK, 1536L * C, wn, ns, NK, PW, acc, plane);
        TSTAMP(33 + 2 * p);
        const int kind = p / PPK, cw = (wave * PPK + p % PPK) * PC;     // PC columns at cw inside the 512-wide Q / K / V
        if (vt_path && kind == 2) {
            // C layout -> [column][frame] patch, one 16-frame fragment at a time: a lane holds 4 consecutive frames of
            // one column; then the 2 * PC 16-byte chunks of 8 frames (column = chunk >> 1) go out one or two per lane
            // (frames >= Tn as zeros: the pad of the transposed buffer stays finite).  Split build: once for the hi
            // terms, once for the remainders.
            char* vw = reinterpret_cast<char*>(patch);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                bf16_t* dst = reinterpret_cast<bf16_t*>(nx.vt_out) + (long)b * nx.vt_bs + (long)s2 * 512 * nx.ldvt + (long)cw * nx.ldvt + t0;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        uint2 pk;
                        pk.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                        pk.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                        *reinterpret_cast<uint2*>(vw + (j * 16 + l16) * PV + (4 * g) * 2) = pk;
                        if (s2 + 1 < NS) {             // the remainders become the next plane
                            acc[i][j][0] -= __uint_as_float(pk.x << 16); acc[i][j][1] -= __uint_as_float(pk.x & 0xffff0000u);
                            acc[i][j][2] -= __uint_as_float(pk.y << 16); acc[i][j][3] -= __uint_as_float(pk.y & 0xffff0000u);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k2 = 0; k2 < PW / 2; ++k2) {
                        const int chunk = lane + 64 * k2, colv = chunk >> 1, c8 = chunk & 1;
                        uint4 v = *reinterpret_cast<const uint4*>(vw + colv * PV + c8 * 16);
                        const int t = t0 + i * 16 + c8 * 8;
                        if (t >= Tn) continue;
                        if (t + 8 > Tn) {
                            unsigned short* h = reinterpret_cast<unsigned short*>(&v);
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (t + e >= Tn) h[e] = 0;
                        }
                        *reinterpret_cast<uint4*>(dst + (long)colv * nx.ldvt + i * 16 + c8 * 8) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else if (planes) {
            bf16_t* out = reinterpret_cast<bf16_t*>(nx.q_out) + (long)b * nx.q_bs;
            const int col = kind * 512 + cw + (lane & 3) * (4 * PW);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float v[4 * PW], lo[4 * PW];
                to_rows<PW>(acc[i], patch, lane, v);
                const int t = t0 + i * 16 + rl;
                if (t < Tn) {
#pragma unroll
                    for (int c = 0; c < 4 * PW; ++c) lo[c] = v[c] - bf2f(f2bf(v[c]));
                    storen_T<bf16_t, 4 * PW>(out + (long)t * nx.ldq + col, v);
                    storen_T<bf16_t, 4 * PW>(out + (long)t * nx.ldq + 1024 + col, lo);
                }
            }
        } else {
            TI* out = reinterpret_cast<TI*>(nx.q_out) + (long)b * nx.q_bs;
            const int col = kind * 512 + cw + (lane & 3) * (4 * PW);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float v[4 * PW];
                to_rows<PW>(acc[i], patch, lane, v);
                const int t = t0 + i * 16 + rl;
                if (t < Tn) storen_T<TI, 4 * PW>(out + (long)t * nx.ldq + col, v);
            }
        }
        TSTAMP(34 + 2 * p);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NW = 4: one wave per SIMD, every wave a 64-column slice of the 256-wide stages.  NW = 8: two waves per SIMD, 32-column
// slices: the VALU-heavy epilogues (GELU, layout changes, LayerNorm) of one wave run under the MFMA stage of the other,
// which a single wave per SIMD cannot do for itself (measured at 64 rows: 60 us = weight stream 23 + MFMA 17 + VALU ~20,
// one after the other).
// OCC = 2 (4-wave workgroups): registers capped at 256 so that TWO workgroups share a CU - no barrier ties them, so one
// workgroup's VALU epilogues run beside the other's weight stream / MFMA stages.
// PW = 2 (8 waves): FF1 and Q/K/V passes of 32 columns, every stage 2 fragments wide - the pass accumulator is 32 instead
// of 64 registers per 64 rows, which is what lets the 64-row tile run two waves per SIMD inside 256 registers without
// spilling (the 64-column-pass version spilled 400 bytes per lane), with a ring 4 k-steps deep: 8 waves x 8 KB of weight
// fragments in flight per CU instead of 4 x 8 KB.  At depth 2 a fragment is requested two k-steps (0.25 us of MFMAs) before
// its use, less than an L2 hit takes, so the one-wave-per-SIMD kernel waits in every k-step (tools/tail_lab.py --stamps).
template <typename T, int BM, int PF, int NW, int NS, int PW, bool WP = false>
__device__ __forceinline__ void est_tail_tile(const MmxEstTailParams& p, const int tile) {
    // HK (the split build's 64-row tile: two bf16 planes of 64 rows): the 512-wide attention tile enters as two K halves of 256 -
    // the first in buf0, the second in a1, which the out projection is done with before LayerNorm writes it - and the FF
    // intermediate passes in four chunks of 256 columns, so that buf0 and a1 are both [2][64][256]: 136 KB + 10 KB of one-fragment
    // patches (to_rows_sp).  Same arithmetic in the same order per row as the 32-row tile: bit-identical results.
    constexpr bool HK = NS == 2 && BM == 64 && sizeof(T) == 2;
    static_assert(!HK || (NW == 8 && PW == 2), "64-row split tile: 8 waves, 32-column passes");
    constexpr int MF = BM / 16, E = FT<T>::E, KB = FT<T>::KB, C = 256, CI = 512, CF = 1024, CH = HK ? 256 : 512, NCH = CF / CH;
    constexpr int WC = C / NW, CW = WC / 4, NFN = WC / 16, PC = 16 * PW, PPC = CH / (PC * NW);
    static_assert(NFN <= PW, "the ring is PW fragments wide");
    constexpr int P0 = tile_pitch(HK ? C : CI, sizeof(T)), P1 = tile_pitch(C, sizeof(T));
    constexpr int PL0 = BM * P0, PL1 = BM * P1;        // bytes between the planes of a tile (split build)
    // GELU: the fp32 build takes libm's erff; the split build erf_fast (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 absolute -
    // two orders below the 2^-17 its products keep; erff was 3.96 us of every FF1 epilogue, 15 % of the kernel:
    // profiles/r04_tail_stamps_x.txt); the bf16 build the degree-17 polynomial
    constexpr bool PRECISE = sizeof(T) == 4;
    typedef std::conditional_t<NS == 1, T, float> TI;   // activation type in HBM
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* buf0 = smem;                                 // [NS][BM][512] attention output, then the FF intermediate chunk
    char* a1 = buf0 + NS * PL0;                        // [NS][BM][256] LayerNorm output (A operand of FF1 / QKV)
    float* patch_all = reinterpret_cast<float*>(a1 + NS * PL1);
    constexpr int PATCHF = HK ? SPATCH_FLOATS : PATCH_FLOATS;
    float* stats = patch_all + NW * PATCHF;            // [BM][NW]
    // the block's bias / LayerNorm vectors: bo | n3g | n3b | b2 | n1g | n1b (256 each) | b1 (1024).  Epilogues read them
    // from here (lgkmcnt) instead of holding them in registers from before the preceding MFMA stage: a global load issued
    // in an epilogue would wait behind the weight ring (vmcnt counts in issue order)
    float* prm = stats + BM * NW;
    constexpr int PRM_BO = 0, PRM_N3G = 256, PRM_N3B = 512, PRM_B2 = 768, PRM_N1G = 1024, PRM_N1B = 1280, PRM_B1 = 1536;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15, rl = lane >> 2;
    float* patch = patch_all + wave * PATCHF;
    const int b = blockIdx.y, t0 = p.t_begin + tile * BM, Tn = p.T;
    const int col0 = wave * WC + (lane & 3) * CW;      // this lane's CW columns of a 256-wide row
    unsigned long long* st = nullptr;
    if constexpr (LAB) {
        st = g_tail_stamps;
        if (st) st += ((long)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 64;
    }
    TSTAMP(0);

    const T* wo = reinterpret_cast<const T*>(p.wo) + (long)lane * E;
    const T* w1 = reinterpret_cast<const T*>(p.w1) + (long)lane * E;
    const T* w2 = reinterpret_cast<const T*>(p.w2) + (long)lane * E;
    constexpr int NK0 = CI / KB, NK1 = C / KB, NK2 = CH / KB, NK2T = CF / KB, NKH = NK0 / 2;
    const long ns0 = (long)NK0 * 64 * E, ns1 = (long)NK1 * 64 * E, ns2 = (long)NK2T * 64 * E;
    WRing<T, PW, PF> ring;
    const T* wo_w = wo + (long)(wave * NFN) * ns0;     // this wave's NFN n-fragments of the 256 output columns

    // Global epilogue operands (residual rows, row mask) are loaded BEFORE the MFMA stage whose epilogue uses them.  A wave
    // waits for a load with s_waitcnt vmcnt(N), which counts in issue order: a load issued in the epilogue would wait for
    // itself AND drain the weight ring that is running ahead for the next stage.  The small vectors come from LDS (prm).
    // ---- attention output projection + bias + residual  (transformer.py:290-297: attn1 -> + hidden_states)
    // FF1 pass q = ch*PPC + h: PC columns at ch*512 + (wave*PPC + h)*PC of the 1024-wide intermediate
    auto w1_pass = [&](int q) { return w1 + (long)(((q / PPC) * CH + (wave * PPC + q % PPC) * PC) / 16) * ns1; };
    float x1[MF][CW];
    float rm[MF];                                      // row mask of the closing epilogue: requested here, with everything else
    {
        const float* rmk = p.rowmask ? p.rowmask + (long)b * p.rm_bs : nullptr;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int t = t0 + i * 16 + rl;
            rm[i] = (rmk && t < Tn) ? rmk[t] : 1.f;
        }
    }
    {
        // Everything the prologue reads from global memory is requested before the first wait (the tile copy's): the
        // parameter vectors into registers, the residual rows (clamped, no branch around a load), the head of the weight
        // ring, then the attention tile - one memory round trip instead of three.
        constexpr int PRM_PER = (TAIL_PRM_FLOATS / 4 + 64 * NW - 1) / (64 * NW);
        float4_t pr[PRM_PER];
#pragma unroll
        for (int k = 0; k < PRM_PER; ++k) {
            const int id0 = tid + k * 64 * NW, id = id0 < TAIL_PRM_FLOATS / 4 ? id0 : TAIL_PRM_FLOATS / 4 - 1;
            const int seg = id >> 6, o4 = (id & 63) * 4;            // 64 float4 per 256-float vector; b1 is segments 6..9
            const float* src = seg == 0 ? p.bo : seg == 1 ? p.n3g : seg == 2 ? p.n3b : seg == 3 ? p.b2
                             : seg == 4 ? (p.next.wqkv ? p.next.n1g : p.b2) : seg == 5 ? (p.next.wqkv ? p.next.n1b : p.b2) : p.b1 + (seg - 6) * 256;
            pr[k] = *reinterpret_cast<const float4_t*>(src + o4);
        }
        const float* xr = p.x + (long)b * p.x_bs;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int t = t0 + i * 16 + rl;
            loadn<CW>(xr + (long)(t < Tn ? t : Tn - 1) * C + col0, x1[i]);
            if (t >= Tn) {
#pragma unroll
                for (int c = 0; c < CW; ++c) x1[i][c] = 0.f;
            }
        }
        ring.prime(wo_w, ns0, NK0, NFN);
        if constexpr (HK) {
            static_assert(P0 == P1 && PL0 == PL1, "both halves of the attention tile have the shape of a1");
            load_tile_split_halves(reinterpret_cast<const float*>(p.ao) + (long)b * p.ao_bs, p.ldao, t0, Tn, BM, buf0, a1, P0, PL0, tid, 64 * NW);
        } else if constexpr (NS == 1) load_tile<T>(reinterpret_cast<const T*>(p.ao) + (long)b * p.ao_bs, p.ldao, t0, Tn, BM, CI, buf0, P0, tid, 64 * NW);
        else load_tile_split(reinterpret_cast<const float*>(p.ao) + (long)b * p.ao_bs, p.ldao, t0, Tn, BM, CI, buf0, P0, PL0, tid, 64 * NW);
#pragma unroll
        for (int k = 0; k < PRM_PER; ++k) {
            const int id = tid + k * 64 * NW;
            if (id < TAIL_PRM_FLOATS / 4) *reinterpret_cast<float4_t*>(prm + id * 4) = pr[k];
        }
        TSTAMP(1);
        __syncthreads();
        TSTAMP(2);
        float4_t acc[MF][NFN];
        zero_acc(acc);
        if constexpr (HK) {                            // k-steps 0 .. 7 over buf0, 8 .. 15 over a1: one accumulator, ascending k
            const T* wo_h = wo_w + (long)NKH * 64 * E;     // second K half of this wave's fragments
            if constexpr (WP) {                        // weight planes: the hi pack over both halves, then the lo pack against the hi activation plane (stage_run_w's order)
                const long wlo = (long)C * CI;
                stage_run<T, MF, NFN, PF, NS, PW>(ring, buf0 + l16 * P0 + g * 16, P0, NKH, wo_w, ns0, NKH, wo_h, ns0, NKH, NFN, acc, PL0);
                stage_run<T, MF, NFN, PF, NS, PW>(ring, a1 + l16 * P1 + g * 16, P1, NKH, wo_h, ns0, NKH, wo_w + wlo, ns0, NKH, NFN, acc, PL1);
                stage_run<T, MF, NFN, PF, 1, PW>(ring, buf0 + l16 * P0 + g * 16, P0, NKH, wo_w + wlo, ns0, NKH, wo_h + wlo, ns0, NKH, NFN, acc, 0);
                stage_run<T, MF, NFN, PF, 1, PW>(ring, a1 + l16 * P1 + g * 16, P1, NKH, wo_h + wlo, ns0, NKH, w1_pass(0), ns1, NK1, PW, acc, 0);
            } else {
                stage_run<T, MF, NFN, PF, NS, PW>(ring, buf0 + l16 * P0 + g * 16, P0, NKH, wo_w, ns0, NKH, wo_h, ns0, NKH, NFN, acc, PL0);
                stage_run<T, MF, NFN, PF, NS, PW>(ring, a1 + l16 * P1 + g * 16, P1, NKH, wo_h, ns0, NKH, w1_pass(0), ns1, NK1, PW, acc, PL1);
            }
        } else
        stage_run_w<T, MF, NFN, PF, NS, PW, WP>(ring, buf0 + l16 * P0 + g * 16, P0, NK0, wo_w, ns0, NK0, (long)C * CI, w1_pass(0), ns1, NK1, PW, acc, PL0);
        TSTAMP(3);
        float bo[CW];
        loadn<CW>(prm + PRM_BO + col0, bo);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v[CW];
            rows_of<NFN, HK>(acc[i], patch, lane, v);
#pragma unroll
            for (int c = 0; c < CW; ++c) x1[i][c] += v[c] + bo[c];
            // (the residual rows stay in registers through both FF stages.  Rounds 3 - 4 parked them in HBM in the narrow-pass 8-wave
            // kernels; reloading them in front of the last FF2 stage cost that stage a memory round trip - vmcnt counts in issue
            // order, the weight ring's next refill waits behind the reload: 3.7 instead of 2.2 us at 64 rows - and without the parking
            // code the same kernels need FEWER registers: 222 / 204 instead of 232 / 218)
        }
        TSTAMP(4);
    }
    // ---- LayerNorm (norm3) -> A1
    float n3g[CW], n3b[CW];
    loadn<CW>(prm + PRM_N3G + col0, n3g);
    loadn<CW>(prm + PRM_N3B + col0, n3b);
    {
        float hn[MF][CW];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int c = 0; c < CW; ++c) hn[i][c] = x1[i][c];
        layernorm_rows<MF, CW, NW>(hn, stats, n3g, n3b, p.eps, wave, lane);
#pragma unroll
        for (int i = 0; i < MF; ++i) store_tile<T, NS, CW>(a1 + (i * 16 + rl) * P1 + col0 * (int)sizeof(T), PL1, hn[i]);
    }
    __syncthreads();                                   // A1 complete; every wave is done with the attention tile
    TSTAMP(5);
    // ---- FF1 + GELU -> LDS chunk -> FF2 accumulate  (transformer.py:306-313, diffusers GELU = Linear + exact gelu)
    float4_t acc2[MF][NFN];
    zero_acc(acc2);
    const T* w2_w = w2 + (long)(wave * NFN) * ns2;     // FF2: this wave's output columns, K walked per chunk
    for (int ch = 0; ch < NCH; ++ch) {
        // (lab stamps) HK: five per chunk from 6 on; otherwise the round-3 numbering
        const int sb = HK ? 6 + ch * 5 : 0;
        for (int h = 0; h < PPC; ++h) {
            const int q = ch * PPC + h;
            const int hc = (wave * PPC + h) * PC + (lane & 3) * (4 * PW);   // column inside the chunk
            float4_t acc[MF][PW];
            zero_acc(acc);
            const bool more = h + 1 < PPC;
            const T* wn = more ? w1_pass(q + 1) : w2_w + (long)(ch * NK2) * 64 * E;
            stage_run_w<T, MF, PW, PF, NS, PW, WP>(ring, a1 + l16 * P1 + g * 16, P1, NK1, w1_pass(q), ns1, NK1, (long)CF * C, wn, more ? ns1 : ns2,
                                                   more ? NK1 : NK2, more ? PW : NFN, acc, PL1);
            TSTAMP(HK ? sb : 6 + ch * 12 + h * 2);
            if constexpr (HK) {
                // 64-row split tile: the epilogue in the MFMA's own layout (lane = column l16 of a fragment, four rows 4g .. 4g + 3) - no
                // patch round trip, every element independent of the others.  Lanes l16 / l16 ^ 1 swap two values (DPP) so that each
                // stores two rows of two adjacent columns as one dword per plane.  The same values as the row-layout form below.
                const bool odd = lane & 1;
                const int cb = (wave * PPC + h) * PC;      // this wave's PC columns inside the chunk
                float bj[PW];
#pragma unroll
                for (int j = 0; j < PW; ++j) bj[j] = prm[PRM_B1 + ch * CH + cb + j * 16 + l16];
                char* dst = buf0 + (4 * g + (odd ? 2 : 0)) * P0 + (cb + (l16 & ~1)) * (int)sizeof(T);
#pragma unroll
                for (int r = 0; r < MF; ++r)
#pragma unroll
                    for (int j = 0; j < PW; ++j) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const f32x2_t gl = gelu_fast2(f32x2_t{acc[r][j][e] + bj[j], acc[r][j][e + 1] + bj[j]});
                            v[e] = gl.x; v[e + 1] = gl.y;
                        }
                        const float y0 = dpp_f32<0xB1>(odd ? v[0] : v[2]), y1 = dpp_f32<0xB1>(odd ? v[1] : v[3]);
                        // even lane: rows 0, 1 = (own column, partner's); odd lane: rows 2, 3 = (partner's column, own)
                        const float a0 = odd ? y0 : v[0], c0 = odd ? v[2] : y0, a1v = odd ? y1 : v[1], c1 = odd ? v[3] : y1;
                        const unsigned h0 = pack_bf16x2(a0, c0), h1 = pack_bf16x2(a1v, c1);
                        const unsigned l0 = pack_bf16x2(a0 - __uint_as_float(h0 << 16), c0 - __uint_as_float(h0 & 0xffff0000u));
                        const unsigned l1 = pack_bf16x2(a1v - __uint_as_float(h1 << 16), c1 - __uint_as_float(h1 & 0xffff0000u));
                        char* q2 = dst + r * 16 * P0 + j * 16 * (int)sizeof(T);
                        *reinterpret_cast<unsigned*>(q2) = h0;
                        *reinterpret_cast<unsigned*>(q2 + P0) = h1;
                        *reinterpret_cast<unsigned*>(q2 + PL0) = l0;
                        *reinterpret_cast<unsigned*>(q2 + PL0 + P0) = l1;
                    }
                TSTAMP(sb + 1);
                continue;
            }
            float b1[4 * PW];
            loadn<4 * PW>(prm + PRM_B1 + ch * CH + hc, b1);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float v[4 * PW];
                rows_of<PW, HK>(acc[i], patch, lane, v);
                if constexpr (NS > 1 && !PRECISE) {    // split build: the spelled-out two-element form (common.h: the same bits whatever
                    // the vectoriser does with the loop around it - the 64- and 32-row tiles must agree bit for bit)
#pragma unroll
                    for (int c = 0; c < 4 * PW; c += 2) {
                        const f32x2_t gl = gelu_fast2(f32x2_t{v[c] + b1[c], v[c + 1] + b1[c + 1]});
                        v[c] = gl.x; v[c + 1] = gl.y;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4 * PW; ++c) v[c] = act_c<((PRECISE || NS > 1) ? ACT_GELU : ACT_GELU_POLY), PRECISE>(v[c] + b1[c], 0.f);
                }
                store_tile<T, NS, 4 * PW>(buf0 + (i * 16 + rl) * P0 + hc * (int)sizeof(T), PL0, v);
            }
            TSTAMP(HK ? sb + 1 : 7 + ch * 12 + h * 2);
        }
        __syncthreads();                               // the chunk is complete
        TSTAMP(HK ? sb + 2 : 14 + ch * 12);
        const T* wn = ch + 1 < NCH ? w1_pass((ch + 1) * PPC) : (p.next.wqkv ? qkv_pass<T, NW, PW>(p.next.wqkv, wave, lane, 0) : nullptr);
        if constexpr (WP && !HK) {
            // weight planes: hi and lo pack alternate per 256 columns of the intermediate (two halves of this 512-wide chunk) - the
            // order in which the 64-row tile, whose chunks are 256 wide, meets them: both tiles sum acc2 alike, bit for bit
            constexpr int NKQ = NK2 / 2;
            const T* wh = w2_w + (long)(ch * NK2) * 64 * E;
            stage_run_w<T, MF, NFN, PF, NS, PW, WP>(ring, buf0 + l16 * P0 + g * 16, P0, NKQ, wh, ns2, NKQ, (long)C * CF, wh + (long)NKQ * 64 * E, ns2, NKQ, NFN, acc2, PL0);
            stage_run_w<T, MF, NFN, PF, NS, PW, WP>(ring, buf0 + NKQ * KB * (int)sizeof(T) + l16 * P0 + g * 16, P0, NKQ, wh + (long)NKQ * 64 * E, ns2, NKQ, (long)C * CF, wn, ns1, NK1, PW, acc2, PL0);
        } else
        stage_run_w<T, MF, NFN, PF, NS, PW, WP>(ring, buf0 + l16 * P0 + g * 16, P0, NK2, w2_w + (long)(ch * NK2) * 64 * E, ns2, NK2, (long)C * CF, wn, ns1, NK1, PW, acc2, PL0);
        TSTAMP(HK ? sb + 3 : 15 + ch * 12);
        __syncthreads();                               // every wave is done reading the chunk
        TSTAMP(HK ? sb + 4 : 16 + ch * 12);
    }
    // ---- + bias + residual -> x (fp32 residual stream, in place)
    // HK: buf0 is dead from here on (every wave is past the last chunk's closing barrier): the wide per-wave patches of the
    // other tiles live there for the closing epilogue and the Q/K/V epilogues (the one-fragment patch serves half the lanes per trip)
    float* wpatch = HK ? reinterpret_cast<float*>(buf0) + wave * PATCH_FLOATS : patch;
    {
        float* xw = p.x + (long)b * p.x_bs;
        float b2[CW];
        loadn<CW>(prm + PRM_B2 + col0, b2);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v[CW];
            to_rows<NFN>(acc2[i], wpatch, lane, v);
            const int t = t0 + i * 16 + rl;
#pragma unroll
            for (int c = 0; c < CW; ++c) x1[i][c] = (x1[i][c] + v[c] + b2[c]) * rm[i];
            if (t < Tn) {
                storen<CW>(xw + (long)t * C + col0, x1[i]);
                if (p.act_out)
                    storen_T<TI, CW>(reinterpret_cast<TI*>(p.act_out) + (long)b * p.act_bs + (long)t * p.act_ld + col0, x1[i]);
            }
        }
    }
    TSTAMP(31);
    if (p.next.wqkv) {
        float n1g[CW], n1b[CW];
        loadn<CW>(prm + PRM_N1G + col0, n1g);
        loadn<CW>(prm + PRM_N1B + col0, n1b);
        ln_qkv<T, MF, PF, NW, NS, PW, WP>(x1, n1g, n1b, p.next, p.eps, a1, wpatch, stats, ring, b, t0, Tn, wave, lane, PL1, st);
    }
    TSTAMP(63);
}

template <typename T, int BM, int PF, int NW, int NS = 1, int OCC = 1, int PW = 4, bool WP = false>
__global__ __launch_bounds__(64 * NW, OCC) void est_tail_kernel(MmxEstTailParams p) {
    // (round 3 / 4 ran the split build's flow groups beside the decode loop with two 32-row tiles per workgroup - half the
    // workgroups per launch; the 64-row split tile does the same at 0.85 of the time and without that form's 180 bytes of scratch)
    est_tail_tile<T, BM, PF, NW, NS, PW, WP>(p, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int PF, int NW, int NS = 1, bool WP = false>
__global__ __launch_bounds__(64 * NW) void est_resnet_kernel(MmxEstResnetParams p) {
    constexpr int MF = BM / 16, MH = MF + 1, E = FT<T>::E, KB = FT<T>::KB, C = 256;
    constexpr int WC = C / NW, CW = WC / 4, NFN = WC / 16;
    constexpr int P1 = tile_pitch(C, sizeof(T));
    constexpr int PLH = (BM + 16) * P1;                // bytes between the planes of h1 (split build)
    constexpr bool PRECISE = sizeof(T) == 4;           // Mish through expf (fp32 build) or v_exp_f32 (1e-6 relative: bf16 and split builds)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cin = p.cin;
    const int PA = tile_pitch(cin, sizeof(T));
    const int PLA = (BM + 18) * PA;                    // bytes between the planes of ain
    char* ain = smem;                                  // [NS][BM + 18][cin]: input rows t0-18 .. t0+BM-1
    char* h1 = ain + NS * PLA;                         // [NS][BM + 16][256]: block1 output rows t0-16 ..; later the A1 tile
    float* patch_all = reinterpret_cast<float*>(h1 + NS * PLH);
    float* stats = patch_all + NW * PATCH_FLOATS;      // [BM + 16][NW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15, rl = lane >> 2;
    float* patch = patch_all + wave * PATCH_FLOATS;
    const int b = blockIdx.y, t0 = p.t_begin + blockIdx.x * BM, Tn = p.T;
    const int col0 = wave * WC + (lane & 3) * CW;
    const int cs = cin / KB;                           // k-steps per tap
    const T* w1 = reinterpret_cast<const T*>(p.w1) + (long)lane * E;
    const T* w2 = reinterpret_cast<const T*>(p.w2) + (long)lane * E;
    const T* wr = reinterpret_cast<const T*>(p.wr) + (long)lane * E;
    const int nk1 = 3 * cs, nk2 = 3 * (C / KB), nkr = cs;
    const long ns1 = (long)nk1 * 64 * E, ns2 = (long)nk2 * 64 * E, nsr = (long)nkr * 64 * E;
    const T* w1_w = w1 + (long)(wave * NFN) * ns1;
    const T* w2_w = w2 + (long)(wave * NFN) * ns2;
    const T* wr_w = wr + (long)(wave * NFN) * nsr;
    WRing<T, 4, PF> ring;
    ring.prime(w1_w, ns1, nk1, NFN);
    const float* rmk = p.rowmask ? p.rowmask + (long)b * p.rm_bs : nullptr;
    // epilogue operands first (see est_tail_kernel): block1's bias, LayerNorm weights, time embedding, row mask
    float bb[CW], tv[CW], gg[CW], be[CW], rm1[MH];
    loadn<CW>(p.b1 + col0, bb);
    loadn<CW>(p.tv + (long)b * p.tv_bs + col0, tv);
    loadn<CW>(p.g1 + col0, gg);
    loadn<CW>(p.be1 + col0, be);
#pragma unroll
    for (int i = 0; i < MH; ++i) {
        const int t = t0 - 16 + i * 16 + rl;
        rm1[i] = (t >= 0 && t < Tn) ? (rmk ? rmk[t] : 1.f) : 0.f;
    }
    if constexpr (NS == 1) load_tile<T>(reinterpret_cast<const T*>(p.a_in) + (long)b * p.a_bs, p.lda, t0 - 18, Tn, BM + 18, cin, ain, PA, tid, 64 * NW);
    else load_tile_split(reinterpret_cast<const float*>(p.a_in) + (long)b * p.a_bs, p.lda, t0 - 18, Tn, BM + 18, cin, ain, PA, PLA, tid, 64 * NW);
    __syncthreads();

    // ---- block1: causal conv k3 (cin -> 256) + bias -> LayerNorm -> Mish -> * mask, + time embedding, * mask
    //      (flow/decoder.py:65-85 with matcha decoder.py:56-61) for rows t0-16 .. t0+BM-1 (conv2 needs 2 rows of halo)
    {
        float4_t acc[MH][NFN];
        zero_acc(acc);
        stage_run_w<T, MH, NFN, PF, NS, 4, WP>(ring, ain + l16 * PA + g * 16, PA, cs, w1_w, ns1, nk1, (long)C * 3 * cin, w2_w, ns2, nk2, NFN, acc, PLA);
        float hv[MH][CW];
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            to_rows<NFN>(acc[i], patch, lane, hv[i]);
#pragma unroll
            for (int c = 0; c < CW; ++c) hv[i][c] += bb[c];
        }
        layernorm_rows<MH, CW, NW>(hv, stats, gg, be, p.eps, wave, lane);
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int t = t0 - 16 + i * 16 + rl;
            float o[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float y = act_c<ACT_MISH, PRECISE>(hv[i][c], 0.f) * rm1[i];
                o[c] = t >= 0 ? (y + tv[c]) * rm1[i] : 0.f;           // rows before the sequence start are conv padding
            }
            store_tile<T, NS, CW>(h1 + (i * 16 + rl) * P1 + col0 * (int)sizeof(T), PLH, o);
        }
    }
    // operands of block2's epilogue and of the residual conv's
    float br[CW];
    loadn<CW>(p.b2 + col0, bb);
    loadn<CW>(p.g2 + col0, gg);
    loadn<CW>(p.be2 + col0, be);
    loadn<CW>(p.br + col0, br);
    __syncthreads();
    // ---- block2: causal conv k3 (256 -> 256) -> LayerNorm -> Mish -> * mask
    float h2[MF][CW];
    {
        float4_t acc[MF][NFN];
        zero_acc(acc);
        stage_run_w<T, MF, NFN, PF, NS, 4, WP>(ring, h1 + (14 + l16) * P1 + g * 16, P1, C / KB, w2_w, ns2, nk2, (long)C * 3 * C, wr_w, nsr, nkr, NFN, acc, PLH);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            to_rows<NFN>(acc[i], patch, lane, h2[i]);
#pragma unroll
            for (int c = 0; c < CW; ++c) h2[i][c] += bb[c];
        }
        layernorm_rows<MF, CW, NW>(h2, stats, gg, be, p.eps, wave, lane);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int c = 0; c < CW; ++c) h2[i][c] = act_c<ACT_MISH, PRECISE>(h2[i][c], 0.f) * rm1[i + 1];
    }
    // ---- + res_conv(x) (1x1) -> x (fp32 residual stream)
    {
        if (p.next.wqkv) { loadn<CW>(p.next.n1g + col0, gg); loadn<CW>(p.next.n1b + col0, be); }
        float4_t acc[MF][NFN];
        zero_acc(acc);
        const T* wq0 = p.next.wqkv ? qkv_pass<T, NW>(p.next.wqkv, wave, lane, 0) : nullptr;
        stage_run_w<T, MF, NFN, PF, NS, 4, WP>(ring, ain + (18 + l16) * PA + g * 16, PA, cs, wr_w, nsr, nkr, (long)C * cin, wq0, (long)(C / KB) * 64 * E, C / KB, 4, acc, PLA);
        float* xw = p.x + (long)b * p.x_bs;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v[CW];
            to_rows<NFN>(acc[i], patch, lane, v);
            const int t = t0 + i * 16 + rl;
#pragma unroll
            for (int c = 0; c < CW; ++c) h2[i][c] += v[c] + br[c];
            if (t < Tn) storen<CW>(xw + (long)t * C + col0, h2[i]);
        }
    }
    __syncthreads();                                   // every wave is done with ain / h1 (h1 is reused by ln_qkv)
    if (p.next.wqkv) ln_qkv<T, MF, PF, NW, NS, 4, WP>(h2, gg, be, p.next, p.eps, h1, patch, stats, ring, b, t0, Tn, wave, lane, PLH);
}


// ---------------------------------------------------------------------------------------------------------------
// DAC-VAE ResidualUnit (dac-vae/model.py:107-143 with :509-514: every Conv1d is followed by LeakyReLU(0.1)) as ONE kernel
// for the narrow stages (C = 48 / 96 / 192 channels, 240 000 .. 40 000 rows per 10 s of audio):
//
//     x_out = x + lrelu(conv1(snake_a2(lrelu(conv7_dil(snake_a0(x))))))          [act_out = snake_next(x_out)]
//
// As two windowed GEMMs (csrc/gemm.hip) a unit moves 16 bytes per element through HBM (activation in, intermediate out
// and in, residual in, residual + next activation out); here the fp32 residual stream is read once (tile + 3 * dilation
// rows of halo each side) and written once.  A workgroup takes BM rows: snake(x) of rows t0 - 3d .. t0 + BM + 3d goes into
// an LDS tile (bf16, or hi + lo planes in the split build), the k7 conv walks it as 7 taps d rows apart, its output
// (+ bias, LeakyReLU, Snake) becomes a second LDS tile, the 1x1 conv reads that, and the epilogue adds the residual.
// Weights are packed in MFMA B-fragment order and streamed from L2 through the register ring of the estimator kernels;
// 4 waves as WR row groups x WC column groups of 48 columns (3 n-fragments) each: C = 192 -> 1 x 4 (every weight
// fragment streamed once per workgroup), 96 -> 2 x 2, 48 -> 4 x 1.  CP = channels padded to a multiple of 32 (48 -> 64,
// zero columns in both tiles, zero weight columns).
// LDS row pitch of a [rows][cp] bf16 tile: the smallest odd multiple of 32 bytes that holds a row (conflict-free
// ds_read_b128 fragment reads, like the 96 / 160-byte pitches of csrc/gemm.hip)
__host__ __device__ constexpr int dac_pitch(int cp) { return (((cp * 2 + 31) / 32) | 1) * 32; }
constexpr int DAC_PATCH_FLOATS = 16 * 52;              // per wave: 48 columns + 4

template <int C, int BM, int WR, int WC, int NS, int PF, bool WP = false>
__global__ __launch_bounds__(256) void dac_ru_kernel(MmxDacRuParams p) {
    typedef bf16_t T;
    constexpr int E = 8, KB = 32, NF = 3, CP = (C + 31) / 32 * 32, PA = dac_pitch(CP);
    constexpr int BMW = BM / WR, MF = BMW / 16;
    constexpr int NK7 = 7 * CP / KB, NK1 = CP / KB;
    constexpr bool PRECISE = NS > 1;
    static_assert(WR * WC == 4 && WC * NF * 16 == C && NK7 % PF == 0 && NK1 % PF == 0, "wave grid / ring depth");
    typedef std::conditional_t<NS == 1, bf16_t, float> TA;   // activation type in HBM
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = p.dil, halo = 3 * d, rows_in = BM + 2 * halo;
    const int PLA = rows_in * PA, PLM = BM * PA;       // bytes between the planes of a tile (split build)
    char* ain = smem;                                  // [NS][BM + 6d][CP] snake_a0(x), rows t0 - 3d ..
    char* mid = ain + NS * PLA;                        // [NS][BM][CP]
    float* patch_all = reinterpret_cast<float*>(mid + NS * PLM);
    float* prm = patch_all + 4 * DAC_PATCH_FLOATS;     // a0 | a2 | b7 | b1 | alpha_next | 1/(a0+1e-9) | 1/(a2+..) | 1/(alpha_next+..)
    // Snake with the reciprocal taken once per channel: the same operations as snake_apply (layers.py:22), in the same order
    auto snk = [](float x, float alpha, float inv) {
        const float sn = PRECISE ? sin_cw(alpha * x) : __sinf(alpha * x);
        return x + inv * (sn * sn);
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15, rl = lane >> 2;
    const int wr = wave / WC, wc = wave % WC;
    float* patch = patch_all + wave * DAC_PATCH_FLOATS;
    const int b = blockIdx.y, t0 = blockIdx.x * BM;
    const int len = p.lens ? p.lens[b] : p.T;          // rows >= len are zero (conv padding of a shorter batch member)
    const int lc = len > 0 ? len : 1;                  // clamp bound for addresses
    const float* xb = p.x + (long)b * p.x_bs;

    const T* w7 = reinterpret_cast<const T*>(p.w7) + (long)lane * E;
    const T* w1 = reinterpret_cast<const T*>(p.w1) + (long)lane * E;
    const long ns7 = (long)NK7 * 64 * E, ns1 = (long)NK1 * 64 * E;
    const T* w7_w = w7 + (long)(wc * NF) * ns7;
    const T* w1_w = w1 + (long)(wc * NF) * ns1;
    WRing<T, NF, PF> ring;
    ring.prime(w7_w, ns7, NK7, NF);

    for (int id = tid; id < 8 * CP; id += 256) {
        const int v = id / CP, c = id - v * CP;
        const float* src = v == 0 ? p.a0 : v == 1 ? p.a2 : v == 2 ? p.b7 : v == 3 ? p.b1 : v == 4 ? p.alpha_next : v == 5 ? p.a0 : v == 6 ? p.a2 : p.alpha_next;
        const float val = (c < C && src) ? src[c] : 0.f;
        prm[id] = v < 5 ? val : 1.0f / (val + 1e-9f);
    }
    __syncthreads();
    // ---- snake_a0(x) of the tile and its halo -> ain.  16-byte chunks of 4 channels; loads in batches of U (no branch around
    //      a load: rows outside [0, len) are clamped and zeroed) - every batch is one HBM round trip for the whole workgroup
    {
        constexpr int CPR = CP / 4, U = 16;
        const int total = rows_in * CPR;
        for (int base = tid; base < total; base += U * 256) {
            float4 v[U];
            int off[U], cch[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = base + u * 256;
                const int idc = id < total ? id : total - 1;
                const int r = idc / CPR, ch = idc - r * CPR;
                const int t = t0 - halo + r;
                const bool ok = t >= 0 && t < len && ch * 4 < C;
                const int tc = t < 0 ? 0 : (t < lc ? t : lc - 1);
                const int cc = ch * 4 < C ? ch * 4 : 0;
                v[u] = *reinterpret_cast<const float4*>(xb + (long)tc * C + cc);
                if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                off[u] = id < total ? r * PA + ch * 8 : -1;
                cch[u] = ch * 4;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (off[u] < 0) continue;
                const float4_t al = *reinterpret_cast<const float4_t*>(prm + cch[u]);     // a0 (zero in the padded columns)
                const float4_t iv = *reinterpret_cast<const float4_t*>(prm + 5 * CP + cch[u]);
                float o[4] = {snk(v[u].x, al[0], iv[0]), snk(v[u].y, al[1], iv[1]), snk(v[u].z, al[2], iv[2]), snk(v[u].w, al[3], iv[3])};
                uint2 hi;
                hi.x = pack_bf16x2(o[0], o[1]);
                hi.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2*>(ain + off[u]) = hi;
                if constexpr (NS > 1) {
                    uint2 lo;
                    lo.x = pack_bf16x2(o[0] - __uint_as_float(hi.x << 16), o[1] - __uint_as_float(hi.x & 0xffff0000u));
                    lo.y = pack_bf16x2(o[2] - __uint_as_float(hi.y << 16), o[3] - __uint_as_float(hi.y & 0xffff0000u));
                    *reinterpret_cast<uint2*>(ain + PLA + off[u]) = lo;
                }
            }
        }
        if constexpr (CP > C) {                        // zero the padded columns of mid (ain's were written as zeros above)
            for (int id = tid; id < BM * (CP - C) / 4; id += 256) {
                const int r = id / ((CP - C) / 4), ch = id % ((CP - C) / 4);
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) *reinterpret_cast<uint2*>(mid + s2 * PLM + r * PA + (C + ch * 4) * 2) = make_uint2(0, 0);
            }
        }
    }
    __syncthreads();
    // ---- conv k7, dilation d: rows wr*BMW .., columns wc*48 ..
    {
        float4_t acc[MF][NF];
        zero_acc(acc);
        stage_run_w<T, MF, NF, PF, NS, NF, WP>(ring, ain + (wr * BMW + l16) * PA + g * 16, PA, CP / KB, w7_w, ns7, NK7, (long)C * 7 * CP, w1_w, ns1, NK1, NF, acc, PLA, d * PA);
        // + bias -> LeakyReLU -> Snake(a2) -> mid (row layout through the wave's patch: 12 consecutive columns per lane)
        const int col = wc * 48 + (lane & 3) * 12;
        float b7[12], a2[12], i2[12];
        loadn<12>(prm + 2 * CP + col, b7);
        loadn<12>(prm + 1 * CP + col, a2);
        loadn<12>(prm + 6 * CP + col, i2);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v[12];
            to_rows<NF>(acc[i], patch, lane, v);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                float y = v[c] + b7[c];
                y = y > 0.f ? y : y * p.slope;
                v[c] = snk(y, a2[c], i2[c]);
            }
            char* dst = mid + (wr * BMW + i * 16 + rl) * PA + col * 2;
            uint2 h0, h1, h2;
            h0.x = pack_bf16x2(v[0], v[1]); h0.y = pack_bf16x2(v[2], v[3]);
            h1.x = pack_bf16x2(v[4], v[5]); h1.y = pack_bf16x2(v[6], v[7]);
            h2.x = pack_bf16x2(v[8], v[9]); h2.y = pack_bf16x2(v[10], v[11]);
            *reinterpret_cast<uint2*>(dst) = h0;
            *reinterpret_cast<uint2*>(dst + 8) = h1;
            *reinterpret_cast<uint2*>(dst + 16) = h2;
            if constexpr (NS > 1) {
                const unsigned w[6] = {h0.x, h0.y, h1.x, h1.y, h2.x, h2.y};
                unsigned l[6];
#pragma unroll
                for (int e = 0; e < 6; ++e)
                    l[e] = pack_bf16x2(v[2 * e] - __uint_as_float(w[e] << 16), v[2 * e + 1] - __uint_as_float(w[e] & 0xffff0000u));
                *reinterpret_cast<uint2*>(dst + PLM) = make_uint2(l[0], l[1]);
                *reinterpret_cast<uint2*>(dst + PLM + 8) = make_uint2(l[2], l[3]);
                *reinterpret_cast<uint2*>(dst + PLM + 16) = make_uint2(l[4], l[5]);
            }
        }
    }
    // residual rows of the closing epilogue (issued before the 1x1 stage: see est_tail_kernel on vmcnt order)
    const int col = wc * 48 + (lane & 3) * 12;
    float xr[MF][12];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int t = t0 + wr * BMW + i * 16 + rl;
        loadn<12>(xb + (long)(t < lc ? t : lc - 1) * C + col, xr[i]);
    }
    __syncthreads();
    // ---- conv k1 + bias -> LeakyReLU -> + x -> x_out [-> Snake(alpha_next) -> act_out]
    {
        float4_t acc[MF][NF];
        zero_acc(acc);
        stage_run_w<T, MF, NF, PF, NS, NF, WP>(ring, mid + (wr * BMW + l16) * PA + g * 16, PA, NK1, w1_w, ns1, NK1, (long)C * CP, (const T*)nullptr, 0, 0, NF, acc, PLM);
        float b1[12], an[12], in[12];
        loadn<12>(prm + 3 * CP + col, b1);
        loadn<12>(prm + 4 * CP + col, an);
        loadn<12>(prm + 7 * CP + col, in);
        float* xo = p.x_out + (long)b * p.x_bs;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v[12];
            to_rows<NF>(acc[i], patch, lane, v);
            const int t = t0 + wr * BMW + i * 16 + rl;
            const bool live = t < len;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                float y = v[c] + b1[c];
                y = y > 0.f ? y : y * p.slope;
                v[c] = live ? y + xr[i][c] : 0.f;
            }
            if (t < p.T) {
                storen<12>(xo + (long)t * C + col, v);
                if (p.act_out) {
#pragma unroll
                    for (int c = 0; c < 12; ++c) v[c] = snk(v[c], an[c], in[c]);
                    TA* ao = reinterpret_cast<TA*>(p.act_out) + (long)b * p.x_bs + (long)t * C + col;
                    if constexpr (NS == 1) {
                        uint2 h0, h1, h2;
                        h0.x = pack_bf16x2(v[0], v[1]); h0.y = pack_bf16x2(v[2], v[3]);
                        h1.x = pack_bf16x2(v[4], v[5]); h1.y = pack_bf16x2(v[6], v[7]);
                        h2.x = pack_bf16x2(v[8], v[9]); h2.y = pack_bf16x2(v[10], v[11]);
                        *reinterpret_cast<uint2*>(ao) = h0;
                        *reinterpret_cast<uint2*>(ao + 4) = h1;
                        *reinterpret_cast<uint2*>(ao + 8) = h2;
                    } else {
                        storen<12>(reinterpret_cast<float*>(ao), v);
                    }
                }
            }
        }
    }
}

template <int C, int BM, int NS>
size_t dac_ru_lds(int dil) {
    constexpr int CP = (C + 31) / 32 * 32, PA = dac_pitch(CP);
    return (size_t)NS * ((size_t)(BM + 6 * dil) * PA + (size_t)BM * PA) + 4 * DAC_PATCH_FLOATS * 4 + 8 * CP * 4;
}

template <typename T, int BM, int NW, int NS = 1>
size_t tail_lds() {
    if (NS == 2 && BM == 64 && sizeof(T) == 2)         // the split build's 64-row tile (HK in est_tail_tile)
        return (size_t)NS * 2 * BM * tile_pitch(256, sizeof(T)) + (size_t)NW * SPATCH_FLOATS * 4 + (size_t)BM * NW * 4 + (size_t)TAIL_PRM_FLOATS * 4;
    return NS * ((size_t)BM * tile_pitch(512, sizeof(T)) + (size_t)BM * tile_pitch(256, sizeof(T))) + (size_t)NW * PATCH_FLOATS * 4 + (size_t)BM * NW * 4 +
           (size_t)TAIL_PRM_FLOATS * 4;
}
template <typename T, int BM, int NW, int NS = 1>
size_t resnet_lds(int cin) {
    return NS * ((size_t)(BM + 18) * tile_pitch(cin, sizeof(T)) + (size_t)(BM + 16) * tile_pitch(256, sizeof(T))) + (size_t)NW * PATCH_FLOATS * 4 +
           (size_t)(BM + 16) * NW * 4;
}

int check_next(const MmxEstNext& nx, int dtype, int T_) {
    if (!nx.wqkv) return MMX_OK;
    MMX_CHECK_ARG(nx.n1g && nx.n1b && nx.q_out);
    MMX_CHECK_ARG(((uintptr_t)nx.q_out % 16) == 0 && nx.q_bs % 8 == 0);
    if (dtype == MMX_X2) {
        if (nx.vt_out) {                               // pre-split planes for mmx_attn_flash_xs
            MMX_CHECK_ARG(nx.ldq % 8 == 0 && nx.ldq >= 2048 && nx.q_bs % 8 == 0 && nx.ldvt % 8 == 0 && nx.ldvt >= ((T_ + 7) / 8) * 8);
            MMX_CHECK_ARG(((uintptr_t)nx.vt_out % 16) == 0 && nx.vt_bs % 8 == 0 && nx.vt_bs >= 2 * 512 * (int64_t)nx.ldvt);
        } else {
            MMX_CHECK_ARG(nx.ldq % 4 == 0 && nx.ldq >= 1536 && nx.q_bs % 4 == 0);
        }
    } else if (dtype == MMX_BF16) {
        MMX_CHECK_ARG(nx.vt_out && nx.ldq % 8 == 0 && nx.ldq >= 1024 && nx.ldvt % 8 == 0 && nx.ldvt >= ((T_ + 7) / 8) * 8);
        MMX_CHECK_ARG(((uintptr_t)nx.vt_out % 16) == 0 && nx.vt_bs % 8 == 0);
    } else {
        MMX_CHECK_ARG(nx.ldq % 4 == 0 && nx.ldq >= 1536);
    }
    return MMX_OK;
}

}  // namespace

#if MMX_LAB
// lab build only (include/mmx_hip_lab.h): buf uint64 [workgroups][waves][64] or NULL (off); set between launches, on the current device
extern "C" int mmx_lab_tail_stamps(void* buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), &p, sizeof(p));
    return e == hipSuccess ? MMX_OK : -(int)e - 1000;
}
#endif

// cfg = pf + 16 * waves: pf = k-steps of weight fragments a wave keeps in flight (2 / 4 / 8, 0 = default for the tile),
// waves = 4 or 8 per workgroup (0 = default)
extern "C" int mmx_est_tail(const MmxEstTailParams* pp, int dtype, int bm, int cfg, hipStream_t stream) {
    MMX_CHECK_ARG(pp != nullptr);
    const MmxEstTailParams& p = *pp;
    const bool wplanes = dtype == MMX_X2W;             // every packed weight (wo, w1, w2, next.wqkv) is [hi pack | lo pack]
    if (wplanes) dtype = MMX_X2;
    MMX_CHECK_ARG(p.ao && p.x && p.wo && p.w1 && p.w2 && p.bo && p.b1 && p.b2 && p.n3g && p.n3b && p.B > 0 && p.T > 0);
    MMX_CHECK_ARG(p.t_begin >= 0 && p.t_begin < p.T && p.t_begin % 16 == 0);
    MMX_CHECK_ARG(p.ldao >= 512 && p.ldao % (dtype == MMX_X2 ? 4 : 8) == 0 && p.ao_bs % (dtype == MMX_X2 ? 4 : 8) == 0 && p.x_bs % 4 == 0);
    MMX_CHECK_ARG(((uintptr_t)p.ao % 16) == 0 && ((uintptr_t)p.x % 16) == 0);
    MMX_CHECK_ARG(!p.act_out || (p.act_ld % (dtype == MMX_X2 ? 4 : 8) == 0 && p.act_bs % (dtype == MMX_X2 ? 4 : 8) == 0 && ((uintptr_t)p.act_out % 16) == 0));
    if (int rc = check_next(p.next, dtype, p.T)) return rc;
    const int nw = (cfg >> 4) & 15, occ2 = (cfg >> 8) & 1;
#define TAILT(TT, BM, PF, NW, NS, OCC, PW, ...)                                                           \
    do {                                                                                                   \
        const size_t lds = tail_lds<TT, BM, NW, NS>();                                                     \
        MMX_CHECK_ARG(lds * OCC <= 160 * 1024);                                                            \
        MMX_LDS_OPT_IN((est_tail_kernel<TT, BM, PF, NW, NS, OCC, PW __VA_OPT__(,) __VA_ARGS__>), lds);     \
        hipLaunchKernelGGL((est_tail_kernel<TT, BM, PF, NW, NS, OCC, PW __VA_OPT__(,) __VA_ARGS__>), dim3((p.T - p.t_begin + BM - 1) / BM, p.B), dim3(64 * NW), lds, stream, p); \
    } while (0)
#define TAILP(TT, BM, PF, NW, NS, OCC, PW) TAILT(TT, BM, PF, NW, NS, OCC, PW)
#define TAILO(TT, BM, PF, NW, NS, OCC) TAILP(TT, BM, PF, NW, NS, OCC, 4)
#define TAILN(TT, BM, PF, NW, NS) TAILO(TT, BM, PF, NW, NS, 1)
#define TAIL(TT, BM, PF, NW) TAILN(TT, BM, PF, NW, 1)
    int narrow = (cfg >> 9) & 1;
    int pf = cfg & 15;
    if (wplanes) {                                     // weight planes: the split build's default tiles
        if (bm == 64) TAILT(bf16_t, 64, 2, 8, 2, 1, 2, true);
        else if (bm == 32) TAILT(bf16_t, 32, 2, 8, 2, 1, 4, true);
        else if (bm == 16) TAILT(bf16_t, 16, 4, 8, 2, 1, 4, true);
        else return MMX_EARG;
        MMX_LAUNCH_CHECK();
        return MMX_OK;
    }
    if ((cfg >> 10) & 1) return MMX_EARG;              // (two row tiles per workgroup: removed, see est_tail_kernel)
    // library defaults of the bf16 build (cfg = 0): the narrow-pass 8-wave kernels for the 64- and 32-row tiles (measured per
    // launch at 10 000 rows: 51.1 us against 58.7 us with 4 waves x 64-column passes; 32 rows, 4 000 rows: 31.8 against 33.0)
    if (cfg == 0 && dtype == MMX_BF16 && (bm == 64 || bm == 32)) { narrow = 1; pf = bm == 64 ? 2 : 8; }
    // split build, 64 rows: two planes of 64 rows fit with the attention tile in K halves and 256-wide FF chunks (est_tail_tile, HK)
    if (dtype == MMX_X2 && bm == 64) { narrow = 1; if (pf != 4) pf = 2; }
    if (occ2) return MMX_EARG;                         // (two 4-wave workgroups per CU: measured slower everywhere, spilled; removed)
    if (narrow) {                                      // 8 waves, 32-column passes (PW = 2)
        if (dtype == MMX_BF16 && bm == 64) TAILP(bf16_t, 64, 2, 8, 1, 1, 2);
        else if (dtype == MMX_BF16 && bm == 32) { if (pf == 8) TAILP(bf16_t, 32, 8, 8, 1, 1, 2); else TAILP(bf16_t, 32, 4, 8, 1, 1, 2); }
        else if (dtype == MMX_X2 && bm == 32) { if (pf == 2) TAILP(bf16_t, 32, 2, 8, 2, 1, 2); else TAILP(bf16_t, 32, 4, 8, 2, 1, 2); }
        else if (dtype == MMX_X2 && bm == 64) { if (pf == 4) TAILP(bf16_t, 64, 4, 8, 2, 1, 2); else TAILP(bf16_t, 64, 2, 8, 2, 1, 2); }
        else return MMX_EARG;
    } else if (dtype == MMX_X2) {
        // split build: two bf16 planes per LDS tile, so the largest tile is 32 rows (137 KB with 8 waves)
        // (a ring 4 k-steps deep measured no faster: 35.7 / 38.5 / 54.6 us at 32 / 128 / 256 workgroups against 37.4 / 40.4 / 55.2 -
        // the stages are bound by the CU's L2 read rate, not by the latency of the loads in flight: profiles/r04_tail_lab_x.txt)
        if (bm == 32) { if (nw == 4) TAILN(bf16_t, 32, 4, 4, 2); else TAILN(bf16_t, 32, 2, 8, 2); }
        // (16 rows: 8 waves by default like the taller tiles - every split tile then sums its LayerNorms alike and an utterance's bits do
        // not depend on the tile height its flow group was given; 4 waves with an explicit cfg)
        else if (bm == 16) { if (nw == 4) TAILN(bf16_t, 16, 8, 4, 2); else TAILN(bf16_t, 16, 4, 8, 2); }
        else return MMX_EARG;
    } else if (dtype == MMX_BF16) {
        if (bm == 64) {                                // (reached with an explicit cfg only: the round-2 / round-3 forms, tools/tail_lab.py)
            if (nw == 0 || nw == 4) { if (pf == 4) TAIL(bf16_t, 64, 4, 4); else TAIL(bf16_t, 64, 2, 4); }
            else return MMX_EARG;
        } else if (bm == 32) {
            if (nw == 8 || nw == 0) { if (pf == 4) TAIL(bf16_t, 32, 4, 8); else TAIL(bf16_t, 32, 2, 8); }
            else if (nw == 4) { if (pf == 2) TAIL(bf16_t, 32, 2, 4); else TAIL(bf16_t, 32, 4, 4); }
            else return MMX_EARG;
        } else if (bm == 16) {
            if (nw == 8) TAIL(bf16_t, 16, 4, 8);
            else if (nw == 0 || nw == 4) TAIL(bf16_t, 16, 8, 4);
            else return MMX_EARG;
        } else return MMX_EARG;
    } else if (dtype == MMX_F32) {
        if (bm == 32 && (nw == 0 || nw == 4)) TAIL(float, 32, 4, 4);
        else if (bm == 16 && (nw == 0 || nw == 4)) TAIL(float, 16, 8, 4);
        else return MMX_EARG;
    } else return MMX_EARG;
#undef TAIL
#undef TAILN
#undef TAILO
#undef TAILP
#undef TAILT
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

extern "C" int mmx_est_resnet(const MmxEstResnetParams* pp, int dtype, int bm, int cfg, hipStream_t stream) {
    MMX_CHECK_ARG(pp != nullptr);
    const MmxEstResnetParams& p = *pp;
    const bool wplanes = dtype == MMX_X2W;             // every packed weight (w1, w2, wr, next.wqkv) is [hi pack | lo pack]
    if (wplanes) dtype = MMX_X2;
    MMX_CHECK_ARG(p.a_in && p.x && p.w1 && p.w2 && p.wr && p.b1 && p.b2 && p.br && p.g1 && p.be1 && p.g2 && p.be2 && p.tv);
    MMX_CHECK_ARG(p.t_begin >= 0 && p.t_begin < p.T && p.t_begin % 16 == 0);
    MMX_CHECK_ARG(p.B > 0 && p.T > 0 && p.cin >= 64 && p.cin % 32 == 0 && p.cin <= 512 && p.lda >= p.cin);
    MMX_CHECK_ARG(p.lda % (dtype == MMX_X2 ? 4 : 8) == 0 && p.a_bs % (dtype == MMX_X2 ? 4 : 8) == 0);
    MMX_CHECK_ARG(((uintptr_t)p.a_in % 16) == 0 && ((uintptr_t)p.x % 16) == 0 && p.x_bs % 4 == 0 && p.tv_bs % 4 == 0 && ((uintptr_t)p.tv % 16) == 0);
    if (int rc = check_next(p.next, dtype, p.T)) return rc;
    const int pf_req = cfg & 15, nw = cfg >> 4;
#define RESNN(TT, BM, PF, NW, NS, ...)                                                                    \
    do {                                                                                                   \
        const size_t lds = resnet_lds<TT, BM, NW, NS>(p.cin);                                              \
        MMX_CHECK_ARG(lds <= 160 * 1024);                                                                  \
        MMX_LDS_OPT_IN((est_resnet_kernel<TT, BM, PF, NW, NS __VA_OPT__(,) __VA_ARGS__>), lds);            \
        hipLaunchKernelGGL((est_resnet_kernel<TT, BM, PF, NW, NS __VA_OPT__(,) __VA_ARGS__>), dim3((p.T - p.t_begin + BM - 1) / BM, p.B), dim3(64 * NW), lds, stream, p); \
    } while (0)
#define RESN(TT, BM, PF, NW) RESNN(TT, BM, PF, NW, 1)
    // ring depth: the deepest of 8 / 4 / 2 the tile height wants that divides every stage's k-step count
    // (conv k3 over cin, conv k3 over 256, 1x1 over cin, Q/K/V over 256); cin = 320 allows 2 (bf16) / 4 (fp32) only
    const int kb = dtype == MMX_F32 ? 16 : 32;
    if (dtype == MMX_X2) {
        // split build (two bf16 planes per LDS tile): 32-row tiles while the input tile leaves room (cin <= 320: 150 KB with
        // 4 waves), 16 rows for the 512-channel input of the up block; ring depth as the bf16 build's rule below
        int pfx = pf_req > 0 ? pf_req : 4;
        while (pfx > 2 && ((3 * p.cin / kb) % pfx || (p.cin / kb) % pfx || (256 / kb) % pfx)) pfx /= 2;
        MMX_CHECK_ARG((p.cin / kb) % 2 == 0);
        if (bm == 32) {
            MMX_CHECK_ARG((resnet_lds<bf16_t, 32, 4, 2>(p.cin)) <= 160 * 1024);
            if (wplanes && nw != 4 && (resnet_lds<bf16_t, 32, 8, 2>(p.cin)) <= 160 * 1024) { if (pfx >= 4) RESNN(bf16_t, 32, 4, 8, 2, true); else RESNN(bf16_t, 32, 2, 8, 2, true); }
            else if (wplanes) { if (pfx >= 4) RESNN(bf16_t, 32, 4, 4, 2, true); else RESNN(bf16_t, 32, 2, 4, 2, true); }
            // 8 waves (two per SIMD: a wave alone on its SIMD issues vector instructions at half the SIMD's rate, and the epilogues
            // here are vector work) where the per-wave patches still fit beside the two-plane tiles: cin = 256, the twelve mid blocks
            else if (nw != 4 && (resnet_lds<bf16_t, 32, 8, 2>(p.cin)) <= 160 * 1024) { if (pfx >= 4) RESNN(bf16_t, 32, 4, 8, 2); else RESNN(bf16_t, 32, 2, 8, 2); }
            else if (pfx >= 4) RESNN(bf16_t, 32, 4, 4, 2); else RESNN(bf16_t, 32, 2, 4, 2);
        } else if (bm == 16) {
            // (8 waves under the same condition as the 32-row tile: both tile heights then sum every LayerNorm alike - bit-identical)
            if (nw != 4 && (resnet_lds<bf16_t, 32, 8, 2>(p.cin)) <= 160 * 1024) {
                if (wplanes) { if (pfx >= 4) RESNN(bf16_t, 16, 4, 8, 2, true); else RESNN(bf16_t, 16, 2, 8, 2, true); }
                else if (pfx >= 4) RESNN(bf16_t, 16, 4, 8, 2); else RESNN(bf16_t, 16, 2, 8, 2);
            }
            else if (wplanes) { if (pfx >= 4) RESNN(bf16_t, 16, 4, 4, 2, true); else RESNN(bf16_t, 16, 2, 4, 2, true); }
            else if (pfx >= 4) RESNN(bf16_t, 16, 4, 4, 2); else RESNN(bf16_t, 16, 2, 4, 2);
        } else return MMX_EARG;
        MMX_LAUNCH_CHECK();
        return MMX_OK;
    }
    // 8 waves (two per SIMD) measured faster for the 64-row ResNet tile (50.6 vs 59.7 us at 14 336 rows); they need 17 KB
    // more LDS for the per-wave patches, which the 512-channel input tile (up block) does not leave
    // (and for the smaller tiles: 31.7 vs 36.2 us at 32 rows, 26.9 vs 29.0 us at 16 rows)
    const size_t lds8 = bm == 64 ? resnet_lds<bf16_t, 64, 8>(p.cin) : (bm == 32 ? resnet_lds<bf16_t, 32, 8>(p.cin) : resnet_lds<bf16_t, 16, 8>(p.cin));
    const bool w8 = dtype == MMX_BF16 && (nw == 8 || (nw == 0 && lds8 <= 160 * 1024));
    const int want = pf_req > 0 ? pf_req : (w8 ? (bm == 64 ? 2 : 4) : (bm == 16 ? 8 : 4));
    MMX_CHECK_ARG(want == 2 || want == 4 || want == 8);
    int pf = want;
    while (pf > 2 && ((3 * p.cin / kb) % pf || (p.cin / kb) % pf || (256 / kb) % pf)) pf /= 2;
    MMX_CHECK_ARG((p.cin / kb) % 2 == 0 && (nw == 0 || nw == 4 || nw == 8));
    if (dtype == MMX_BF16) {
        if (w8) {
            if (bm == 64) RESN(bf16_t, 64, 2, 8);
            else if (bm == 32) { if (pf >= 4) RESN(bf16_t, 32, 4, 8); else RESN(bf16_t, 32, 2, 8); }
            else if (bm == 16) { if (pf >= 4) RESN(bf16_t, 16, 4, 8); else RESN(bf16_t, 16, 2, 8); }
            else return MMX_EARG;
        } else {
            if (bm == 64) { if (pf >= 4) RESN(bf16_t, 64, 4, 4); else RESN(bf16_t, 64, 2, 4); }
            else if (bm == 32) { if (pf >= 4) RESN(bf16_t, 32, 4, 4); else RESN(bf16_t, 32, 2, 4); }
            else if (bm == 16) { if (pf == 8) RESN(bf16_t, 16, 8, 4); else if (pf == 4) RESN(bf16_t, 16, 4, 4); else RESN(bf16_t, 16, 2, 4); }
            else return MMX_EARG;
        }
    } else if (dtype == MMX_F32) {
        if (bm == 16 && (nw == 0 || nw == 4)) { if (pf == 8) RESN(float, 16, 8, 4); else if (pf == 4) RESN(float, 16, 4, 4); else RESN(float, 16, 2, 4); }
        else return MMX_EARG;
    } else return MMX_EARG;
#undef RESN
#undef RESNN
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

extern "C" int mmx_dac_ru(const MmxDacRuParams* pp, int dtype, int bm, hipStream_t stream) {
    MMX_CHECK_ARG(pp != nullptr);
    const MmxDacRuParams& p = *pp;
    MMX_CHECK_ARG(p.x && p.x_out && p.x != p.x_out && p.w7 && p.w1 && p.b7 && p.b1 && p.a0 && p.a2 && p.B > 0 && p.T > 0);
    MMX_CHECK_ARG(p.dil >= 1 && p.dil <= 9 && (p.C == 48 || p.C == 96 || p.C == 192) && p.x_bs % 4 == 0 && p.x_bs >= (int64_t)p.T * p.C);
    MMX_CHECK_ARG(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.x_out % 16) == 0 && ((uintptr_t)p.act_out % 16) == 0 && ((uintptr_t)p.a0 % 16) == 0);
    MMX_CHECK_ARG(!p.act_out || p.alpha_next);
    const bool wplanes = dtype == MMX_X2W;             // w7 / w1 are [hi pack | lo pack]
    if (wplanes) dtype = MMX_X2;
    MMX_CHECK_ARG(dtype == MMX_BF16 || dtype == MMX_X2);
#define DACRU(C_, BM_, WR_, WC_, NS_, PF_, ...)                                                            \
    do {                                                                                                   \
        const size_t lds = dac_ru_lds<C_, BM_, NS_>(p.dil);                                                \
        MMX_CHECK_ARG(lds <= 160 * 1024);                                                                  \
        MMX_LDS_OPT_IN((dac_ru_kernel<C_, BM_, WR_, WC_, NS_, PF_ __VA_OPT__(,) __VA_ARGS__>), lds);       \
        hipLaunchKernelGGL((dac_ru_kernel<C_, BM_, WR_, WC_, NS_, PF_ __VA_OPT__(,) __VA_ARGS__>), dim3((p.T + BM_ - 1) / BM_, p.B), dim3(256), lds, stream, p); \
    } while (0)
    if (wplanes) {                                     // weight planes: the split build's default tile per stage
        if (p.C == 48) DACRU(48, 64, 4, 1, 2, 2, true);
        else if (p.C == 96) { if (p.dil > 3) DACRU(96, 128, 2, 2, 2, 3, true); else DACRU(96, 64, 2, 2, 2, 3, true); }
        else DACRU(192, 32, 1, 4, 2, 2, true);
        MMX_LAUNCH_CHECK();
        return MMX_OK;
    }
    // tile heights: default = the measured best of tools/dac_lab.py for (C, dtype); smaller tiles let two workgroups share a
    // CU (LDS, 256 registers), which overlaps one's load / epilogue phases with the other's MFMA stages
    if (dtype == MMX_BF16) {
        if (p.C == 48) { if (bm == 256) DACRU(48, 256, 4, 1, 1, 2); else if (bm == 0 || bm == 128) DACRU(48, 128, 4, 1, 1, 2); else if (bm == 64) DACRU(48, 64, 4, 1, 1, 2); else return MMX_EARG; }
        else if (p.C == 96) { if (bm == 0 || bm == 256) DACRU(96, 256, 2, 2, 1, 3); else if (bm == 128) DACRU(96, 128, 2, 2, 1, 3); else if (bm == 64) DACRU(96, 64, 2, 2, 1, 3); else return MMX_EARG; }
        else { if (bm == 0) bm = p.dil > 3 ? 32 : 64; if (bm == 128) DACRU(192, 128, 1, 4, 1, 2); else if (bm == 64) DACRU(192, 64, 1, 4, 1, 2); else if (bm == 32) DACRU(192, 32, 1, 4, 1, 2); else return MMX_EARG; }
    } else {
        if (p.C == 48) { if (bm == 128) DACRU(48, 128, 4, 1, 2, 2); else if (bm == 0 || bm == 64) DACRU(48, 64, 4, 1, 2, 2); else return MMX_EARG; }
        else if (p.C == 96) { if (bm == 0) bm = p.dil > 3 ? 128 : 64; if (bm == 128) DACRU(96, 128, 2, 2, 2, 3); else if (bm == 64) DACRU(96, 64, 2, 2, 2, 3); else if (bm == 32) DACRU(96, 32, 2, 2, 2, 3); else return MMX_EARG; }
        else { if (bm == 0 || bm == 32) DACRU(192, 32, 1, 4, 2, 2); else if (bm == 16) DACRU(192, 16, 1, 4, 2, 2); else return MMX_EARG; }
    }
#undef DACRU
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
