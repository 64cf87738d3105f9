// Attention kernels of the flow decoder (see include/mmx_hip.h for the contracts).
//   attn_dense_kernel : general fp32-accumulating kernel (any T, optional ESPnet rel-pos term, key / chunk
//                       masks) — the parity path and the conformer encoder (10 layers, once per utterance).
//   attn_flash_kernel : bf16 MFMA flash attention for the estimator's 56 blocks x 10 Euler steps
//                       (the flow's dominant FLOPs: SURVEY.md §8d, 57.3 of 189.5 GFLOP per call).
#include "common.h"
#include "../../include/mmx_hip.h"
#include <type_traits>
#include <utility>

__device__ __forceinline__ bool key_visible(int i, int j, int Tk, const float* km, int chunk) {
    if (j >= Tk) return false;
    if (km && km[j] == 0.f) return false;
    if (chunk > 0 && j >= (i / chunk + 1) * chunk) return false;
    return true;
}

// ------------------------------------------------------------------------------------------ dense
// block = (b, h, 8 queries); scores live in LDS ([8][Tk] fp32), so Tk <= ~4500.
template <typename T>
__global__ __launch_bounds__(256) void attn_dense_kernel(
    const T* __restrict__ q, long ldq, long q_bs, const T* __restrict__ k, long ldk, long k_bs,
    const T* __restrict__ v, long ldv, long v_bs, T* __restrict__ out, long ldo, long o_bs,
    int H, int Tq, int Tk, float scale, const float* __restrict__ keymask, long km_bs, int chunk,
    const T* __restrict__ pos, long ldp, const float* __restrict__ pos_u, const float* __restrict__ pos_v, int hs,
    int q_begin) {
    constexpr int D = 64, QT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qu = reinterpret_cast<float*>(smem);        // [QT][D]  q (+u)
    float* qv = qu + QT * D;                           // [QT][D]  q + v (rel-pos only)
    float* inv_l = qv + QT * D;                        // [QT]
    float* S = inv_l + QT;                             // [QT][Tk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = q_begin + blockIdx.x * QT;
    q += (long)b * q_bs + h * hs;                     // hs: column stride between heads of q/k/v (D, or 3D for the
    k += (long)b * k_bs + h * hs;                     // head-interleaved qkv of arch_util.QKVAttentionLegacy)
    v += (long)b * v_bs + h * hs;
    out += (long)b * o_bs + h * D;
    const float* km = keymask ? keymask + (long)b * km_bs : nullptr;

    for (int i = tid; i < QT * D; i += 256) {
        int qi = i / D, d = i % D;
        int row = q0 + qi < Tq ? q0 + qi : Tq - 1;
        float x = Cvt<T>::to_f(q[(long)row * ldq + d]);
        qu[i] = x + (pos ? pos_u[h * D + d] : 0.f);
        qv[i] = x + (pos ? pos_v[h * D + d] : 0.f);
    }
    __syncthreads();
    // phase 1: scores
    for (int j = tid; j < Tk; j += 256) {
        float kr[D];
        const T* kp = k + (long)j * ldk;
#pragma unroll
        for (int d = 0; d < D; ++d) kr[d] = Cvt<T>::to_f(kp[d]);
#pragma unroll 1
        for (int qi = 0; qi < QT; ++qi) {
            const int i = q0 + qi;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s += qu[qi * D + d] * kr[d];
            if (pos && i < Tq) {
                // rel_shift folded: bd[i][j] = (q_i + v) . pos[Tq - 1 - i + j]
                const T* pp = pos + (long)(Tq - 1 - i + j) * ldp + h * D;
                float s2 = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) s2 += qv[qi * D + d] * Cvt<T>::to_f(pp[d]);
                s += s2;
            }
            S[(long)qi * Tk + j] = key_visible(i, j, Tk, km, chunk) ? s * scale : -INFINITY;
        }
    }
    __syncthreads();
    // phase 2: softmax per row (wave per row)
    for (int qi = wave; qi < QT; qi += 4) {
        float m = -INFINITY;
        for (int j = lane; j < Tk; j += 64) m = fmaxf(m, S[(long)qi * Tk + j]);
        m = wave_max(m);
        float l = 0.f;
        if (m > -INFINITY) {
            for (int j = lane; j < Tk; j += 64) {
                float e = expf(S[(long)qi * Tk + j] - m);
                S[(long)qi * Tk + j] = e;
                l += e;
            }
            l = wave_sum(l);
        } else {
            for (int j = lane; j < Tk; j += 64) S[(long)qi * Tk + j] = 0.f;
        }
        if (lane == 0) inv_l[qi] = l > 0.f ? 1.f / l : 0.f;
    }
    __syncthreads();
    // phase 3: O = P V ; thread -> (query, 2 consecutive d)
    {
        const int qi = tid >> 5, d0 = (tid & 31) * 2;
        float a0 = 0.f, a1 = 0.f;
        const float* Sr = S + (long)qi * Tk;
        for (int j = 0; j < Tk; ++j) {
            float p = Sr[j];
            const T* vp = v + (long)j * ldv + d0;
            a0 += p * Cvt<T>::to_f(vp[0]);
            a1 += p * Cvt<T>::to_f(vp[1]);
        }
        const int i = q0 + qi;
        if (i < Tq) {
            out[(long)i * ldo + d0] = Cvt<T>::from_f(a0 * inv_l[qi]);
            out[(long)i * ldo + d0 + 1] = Cvt<T>::from_f(a1 * inv_l[qi]);
        }
    }
}

extern "C" int mmx_attn_dense(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                              const void* v, int64_t ldv, int64_t v_bs, void* out, int64_t ldo, int64_t o_bs,
                              int B, int H, int D, int Tq, int Tk, float scale, const float* keymask, int64_t km_bs,
                              int chunk, const void* pos, int64_t ldp, const float* pos_u, const float* pos_v,
                              int head_stride, int q_begin, int dtype, hipStream_t stream) {
    dtype = MMX_ACT_DTYPE(dtype);
    MMX_CHECK_ARG(q && k && v && out && B > 0 && H > 0 && D == 64 && Tq > 0 && Tk > 0 && chunk >= 0);
    MMX_CHECK_ARG(q_begin >= 0 && q_begin < Tq);
    const int hs = head_stride > 0 ? head_stride : D;
    MMX_CHECK_ARG(!pos || (pos_u && pos_v && Tq == Tk));
    size_t lds = (size_t)(2 * 8 * 64 + 8 + 8 * (size_t)Tk) * 4;
    MMX_CHECK_ARG(lds <= 160 * 1024);
    dim3 grid((Tq - q_begin + 7) / 8, H, B);
    if (dtype == MMX_BF16)
        hipLaunchKernelGGL(attn_dense_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs,
                           (const bf16_t*)v, ldv, v_bs, (bf16_t*)out, ldo, o_bs, H, Tq, Tk, scale, keymask, km_bs, chunk,
                           (const bf16_t*)pos, ldp, pos_u, pos_v, hs, q_begin);
    else if (dtype == MMX_F32)
        hipLaunchKernelGGL(attn_dense_kernel<float>, grid, dim3(256), lds, stream, (const float*)q, ldq, q_bs, (const float*)k, ldk, k_bs,
                           (const float*)v, ldv, v_bs, (float*)out, ldo, o_bs, H, Tq, Tk, scale, keymask, km_bs, chunk,
                           (const float*)pos, ldp, pos_u, pos_v, hs, q_begin);
    else return MMX_EARG;
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

// ------------------------------------------------------------------------------------------ flash (bf16, D = 64)
// block = 4 waves x 32 queries (2 MFMA row fragments per wave, so every K / V^T fragment read from LDS feeds two
// MFMAs); K tile [64 keys][64 d] and V^T tile [64 d][64 keys] double-buffered in LDS with 144-byte rows
// (72 bf16: 16 rows x 144 B hit 16 distinct 16-byte bank slots -> conflict-free ds_read_b128), register-staged
// prefetch two tiles ahead, ONE workgroup barrier per key tile; S = Q K^T and O += P V on
// v_mfma_f32_16x16x32_bf16; online softmax in registers (exp2 with log2e folded into the scale), row reductions
// over the 16 lanes that share a query row via 4 xor-shuffles; P goes through a per-wave LDS patch (wave-local
// ordering only) to turn the C-layout tile into the A-operand layout.
// FP8 (BASELINE config 5 "fp8 MFMA attention"): Q, K, V^T and P (scaled by 4) are quantised to OCP e4m3 on the fly (bf16 in
// HBM either way) and both products run on v_mfma_f32_16x16x32_fp8_fp8: half the LDS bytes per tile and per
// fragment read.  On gfx950 the non-scaled fp8 MFMA has the bf16 rate and a 64-wide head cannot fill the K = 128 of the
// block-scaled form, so the gain is LDS traffic, not matrix throughput; the price is 3-bit mantissas.
__device__ __forceinline__ unsigned pk_fp8x4(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}
__device__ __forceinline__ uint2 bf16x8_to_fp8(uint4 v) {      // 8 bf16 -> 8 e4m3 bytes
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    return make_uint2(pk_fp8x4(f[0], f[1], f[2], f[3]), pk_fp8x4(f[4], f[5], f[6], f[7]));
}

// NW waves of 16 * MF queries.  NW = 8, MF = 1 (two waves per SIMD, 128 queries per workgroup as with NW = 4, MF = 2): the softmax
// VALU phase of one wave runs under the MFMA phase of its SIMD partner (one wave per SIMD runs them one after the other: PMC
// of round 2, VALU active 50 %, MFMA busy 17 %); every K / V^T fragment read then feeds one MFMA instead of two.
template <int MF, bool FP8, int NW = 4>
__global__ __launch_bounds__(64 * NW) void attn_flash_kernel(
    const bf16_t* __restrict__ q, long ldq, long q_bs, const bf16_t* __restrict__ k, long ldk, long k_bs,
    const bf16_t* __restrict__ vt, long ldvt, long vt_bs, bf16_t* __restrict__ out, long ldo, long o_bs,
    int Tn, float scale, const float* __restrict__ keymask, long km_bs, int chunk, int nq, int nheads, int npairs,
    int q_begin, const int32_t* __restrict__ klen) {
    // LDS row pitches.  A fragment read is ds_read_b128 at (row l16, 16-byte chunk g); the hardware serves it in the lane
    // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. 16 different rows with two
    // adjacent chunks per group: a 144 B pitch puts 7 of the 16 lanes on busy banks (8 LDS cycles instead of 4; PMC:
    // 39 % of this kernel's LDS cycles were bank conflicts), 160 B is conflict free.  The P patch keeps 144 B: its
    // 8-byte stores (16 contiguous lanes, 32 banks) would be 4-way conflicted at 160 B.
    // MF query fragments of 16 per wave: 2 for long / batched problems (every K / V^T fragment read feeds two MFMAs),
    // 1 when the grid would otherwise leave most CUs idle (one 10 s utterance: 16 (batch, head) pairs x 4 tiles of 128)
    constexpr int D = 64, KT = 64, LDK = 80, LD = 72, QW = 16 * MF;
    constexpr int LD8 = 72;                            // byte pitch of the fp8 tiles and of the fp8 P patch
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][KT * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][D * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Ps[NW][QW * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    // XCD-aware mapping (1-D grid): the query tiles of one (batch, head) pair read the same K / V^T rows; they get
    // linear ids with the same id % 8, i.e. the same XCD and L2, instead of being dealt round robin over all eight.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd;
    if (pair >= npairs) return;                        // uniform: the grid is padded to a multiple of 8 pairs
    const int qt = slot % nq;
    const int b = pair / nheads, h = pair % nheads;
    const int qb = q_begin + qt * (NW * QW) + wave * QW;   // this wave's first query
    q += (long)b * q_bs + h * D;
    k += (long)b * k_bs + h * D;
    vt += (long)b * vt_bs + (long)h * D * ldvt;
    out += (long)b * o_bs + h * D;
    const float* km = keymask ? keymask + (long)b * km_bs : nullptr;
    const float sc2 = scale * 1.44269504088896341f;    // scores in log2 units
    // klen: the batch row's number of valid keys (a padded batch whose masks are prefixes).  Unlike a key mask it is known
    // before the loop: key tiles beyond it are never visited, tiles inside it run the unmasked code, only the boundary
    // tile compares; a workgroup whose queries are all padding writes zeros and leaves.
    const int Tk = klen ? (klen[b] < Tn ? klen[b] : Tn) : Tn;
    if (klen && q_begin + qt * (NW * QW) >= Tk) {        // uniform per workgroup, before any barrier
        for (int id = tid; id < NW * QW * 8; id += 64 * NW) {
            const int i = q_begin + qt * (NW * QW) + (id >> 3);
            if (i < Tn) *reinterpret_cast<uint4*>(out + (long)i * ldo + (id & 7) * 8) = make_uint4(0, 0, 0, 0);
        }
        return;
    }

    // Everything is computed TRANSPOSED so that a lane owns ONE query: S^T = K Q^T (MFMA A = K fragment,
    // B = Q fragment) puts query l16 on the lane and keys 4g+r in its registers, so the softmax row reductions are
    // in-lane plus two xor-shuffles (16, 32), m / l / alpha are per-lane scalars, and 4 consecutive keys of P pack
    // into one 8-byte LDS write.  O^T = V^T P^T (A = V^T fragment, B = P fragment read like an A operand) keeps the
    // query on the lane for the rescale and yields 4 consecutive channels per lane for 8-byte output stores.
    short8_t aq[MF][2];                                // lane (q = l16, k-group g): Q[q][ks*32 + 8g .. +7]
    long aq8[MF][2];                                   // the same as 8 e4m3 bytes
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int row = qb + mf * 16 + l16;
        row = row < Tn ? row : Tn - 1;
        const bf16_t* qp = q + (long)row * ldq + 8 * g;
        aq[mf][0] = *reinterpret_cast<const short8_t*>(qp);
        aq[mf][1] = *reinterpret_cast<const short8_t*>(qp + 32);
        if constexpr (FP8) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) aq8[mf][ks] = __builtin_bit_cast(long, bf16x8_to_fp8(__builtin_bit_cast(uint4, aq[mf][ks])));
        }
    }
    float4_t o[MF][4];                                 // O^T: [mf][df] rows d = df*16 + 4g + r, column q = l16
    float m_run[MF], l_run[MF];
    int lim[MF];                                       // keys j < lim are visible to this lane's query
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[mf][i] = float4_t{0.f, 0.f, 0.f, 0.f};
        m_run[mf] = -INFINITY;
        l_run[mf] = 0.f;
        const int i = qb + mf * 16 + l16;
        int e = Tk;
        if (chunk > 0) { int c2 = (i / chunk + 1) * chunk; e = c2 < e ? c2 : e; }
        lim[mf] = e;
    }
    // keys beyond the last query's chunk are invisible to the whole block
    int kend = Tk;
    if (chunk > 0) {
        int qlast = q_begin + qt * (NW * QW) + NW * QW - 1;
        if (qlast > Tn - 1) qlast = Tn - 1;
        int e = (qlast / chunk + 1) * chunk;
        if (e < kend) kend = e;
    }
    const int ntile = (kend + KT - 1) / KT;
    // keys below vis_all are visible to EVERY query of the workgroup (its first query's chunk end): those tiles run the
    // unmasked code; only the tiles that reach into the block's own chunks compare
    int vis_all = Tk;
    if (chunk > 0) {
        const int e = ((q_begin + qt * (NW * QW)) / chunk + 1) * chunk;
        if (e < vis_all) vis_all = e;
    }
    bf16_t* Pw = Ps[wave];

    uint4 kreg[2], vreg[2];
    auto load_tiles = [&](int j0) {
#pragma unroll
        for (int i = 0; i < 512 / (64 * NW); ++i) {
            int id = tid + i * 64 * NW;                // 512 chunks of 16 B per tile
            int r = id >> 3, c = (id & 7) * 8;
            int key = j0 + r;
            kreg[i] = key < Tk ? *reinterpret_cast<const uint4*>(k + (long)key * ldk + c) : make_uint4(0, 0, 0, 0);
            // V^T rows are d; columns j0 + c .. +7 (buffer is zero padded to a multiple of 8 columns)
            vreg[i] = (j0 + c < Tk) ? *reinterpret_cast<const uint4*>(vt + (long)r * ldvt + j0 + c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 512 / (64 * NW); ++i) {
            int id = tid + i * 64 * NW;
            int r = id >> 3, c = (id & 7) * 8;
            if constexpr (FP8) {
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Ks[buf]) + r * LD8 + c) = bf16x8_to_fp8(kreg[i]);
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Vs[buf]) + r * LD8 + c) = bf16x8_to_fp8(vreg[i]);
            } else {
                *reinterpret_cast<uint4*>(Ks[buf] + r * LDK + c) = kreg[i];
                *reinterpret_cast<uint4*>(Vs[buf] + r * LDK + c) = vreg[i];
            }
        }
    };
    load_tiles(0);
    store_tiles(0);
    if (ntile > 1) load_tiles(KT);
    for (int jt = 0; jt < ntile; ++jt) {
        const int j0 = jt * KT, buf = jt & 1;
        __syncthreads();                               // tile jt is in LDS; every wave is done with tile jt-1
        if (jt + 1 < ntile) store_tiles(buf ^ 1);      // tile jt+1 (its buffer was last read for tile jt-1)
        if (jt + 2 < ntile) load_tiles(j0 + 2 * KT);
        // S^T = K Q^T: s[mf][nf][r] = score(query qb + mf*16 + l16, key j0 + nf*16 + 4g + r)
        float4_t s[MF][4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) s[mf][nf] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (FP8) {
                    const long bk = *reinterpret_cast<const long*>(reinterpret_cast<const char*>(Ks[buf]) + (nf * 16 + l16) * LD8 + ks * 32 + 8 * g);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
                        s[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bk, aq8[mf][ks], s[mf][nf], 0, 0, 0);
                } else {
                    short8_t bk = *reinterpret_cast<const short8_t*>(Ks[buf] + (nf * 16 + l16) * LDK + ks * 32 + 8 * g);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
                        s[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk, aq[mf][ks], s[mf][nf], 0, 0, 0);
                }
            }
        }
        const bool need_mask = km || (j0 + KT > vis_all);               // uniform per tile
        bool kvis[4][4];                               // key-side visibility of this lane's 16 keys, once per tile
        if (need_mask) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + nf * 16 + 4 * g + r;
                    kvis[nf][r] = j < Tk && (!km || km[j] != 0.f);
                }
        }
        // softmax of one query fragment; MASK is a compile-time flag so interior tiles carry no compare/select code
        auto softmax_tile = [&](auto mask_c, auto mf_c) {
            constexpr bool MASK = decltype(mask_c)::value;
            constexpr int mf = decltype(mf_c)::value;
            // the scale (log2 units, > 0) is not applied to the scores: max(s) * sc2 == max(s * sc2), and the exponent
            // below is one fma, exp2(s * sc2 - m); saves a multiply per score in a VALU-bound loop
            float mx = -INFINITY;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = s[mf][nf][r];
                    if constexpr (MASK) {
                        const int j = j0 + nf * 16 + 4 * g + r;
                        x = (j < lim[mf] && kvis[nf][r]) ? x : -INFINITY;
                        s[mf][nf][r] = x;
                    }
                    mx = fmaxf(mx, x);
                }
            mx *= sc2;                                 // this LANE's maximum: keys 4g + r of the tile's four 16-key fragments
            // lazy rescale (cdna_hip_programming.md T13): keep the old running max while every score of the wave is at
            // most 2^6 above it; p then reaches at most 64 (fine in bf16 / fp32 sums) and the O accumulators (AGPRs: a
            // rescale costs a read + multiply + write per value) are left alone.  The test needs no cross-lane maximum
            // (each lane checks its own keys against the query's running max, one wave vote), so a steady-state tile has
            // no shuffle at all (ds_bpermute round trips sat on the critical path of a one-wave-per-SIMD loop); the four
            // lanes of a query agree on the maximum only when it moves.  m_run stays uniform over those four lanes.
            float m_use = m_run[mf];
            const bool grow = (mx - m_run[mf]) > 6.0f || m_run[mf] == -INFINITY;
            if (__any(grow)) {
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[mf], mx);
                const float m_safe = m_new == -INFINITY ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[mf] - m_safe);   // m_run = -inf -> 0
                l_run[mf] *= alpha;
#pragma unroll
                for (int df = 0; df < 4; ++df)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[mf][df][r] *= alpha;
                m_run[mf] = m_new;
                m_use = m_safe;
            }
            float rs = 0.f;
            // FP8: P is stored as e4m3 scaled by 4 (values reach 64 under the lazy rescale, e4m3 holds 448; the small end
            // then goes down to 2^-11) - the scale cancels because the denominator sums the same scaled values
            const float m_exp = FP8 ? m_use - 2.0f : m_use;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mf][nf][0], sc2, -m_exp));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mf][nf][1], sc2, -m_exp));
                float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mf][nf][2], sc2, -m_exp));
                float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mf][nf][3], sc2, -m_exp));
                // 4 consecutive keys of query l16 -> one 8-byte write into the row-major [q][key] patch
                if constexpr (FP8) {
                    // the denominator sums the ROUNDED probabilities (what the PV product multiplies): a 3-bit mantissa
                    // rounds a row of similar values all the same way, and an unrounded sum would leave that as a bias
                    const unsigned pk8 = pk_fp8x4(p0, p1, p2, p3);
                    rs += (__builtin_amdgcn_cvt_f32_fp8(pk8, 0) + __builtin_amdgcn_cvt_f32_fp8(pk8, 1)) +
                          (__builtin_amdgcn_cvt_f32_fp8(pk8, 2) + __builtin_amdgcn_cvt_f32_fp8(pk8, 3));
                    *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(Pw) + (mf * 16 + l16) * LD8 + nf * 16 + 4 * g) = pk8;
                } else {
                    rs += (p0 + p1) + (p2 + p3);
                    uint2 pk;
                    pk.x = pack_bf16x2(p0, p1);
                    pk.y = pack_bf16x2(p2, p3);
                    *reinterpret_cast<uint2*>(Pw + (mf * 16 + l16) * LD + nf * 16 + 4 * g) = pk;
                }
            }
            l_run[mf] += rs;                           // per-lane partial sum (its own keys); the four lanes of a query are
        };                                             // added up once, after the last tile
        if (need_mask) {
            softmax_tile(std::true_type{}, std::integral_constant<int, 0>{});
            if constexpr (MF > 1) softmax_tile(std::true_type{}, std::integral_constant<int, MF - 1>{});
        } else {
            softmax_tile(std::false_type{}, std::integral_constant<int, 0>{});
            if constexpr (MF > 1) softmax_tile(std::false_type{}, std::integral_constant<int, MF - 1>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (FP8) {
            long ap8[MF][2];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    ap8[mf][ks] = *reinterpret_cast<const long*>(reinterpret_cast<const char*>(Pw) + (mf * 16 + l16) * LD8 + ks * 32 + 8 * g);
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const long bv = *reinterpret_cast<const long*>(reinterpret_cast<const char*>(Vs[buf]) + (df * 16 + l16) * LD8 + ks * 32 + 8 * g);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
                        o[mf][df] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bv, ap8[mf][ks], o[mf][df], 0, 0, 0);
                }
        } else {
        short8_t ap[MF][2];                            // lane (q = l16, g): P[q][ks*32 + 8g .. +7]
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            ap[mf][0] = *reinterpret_cast<const short8_t*>(Pw + (mf * 16 + l16) * LD + 8 * g);
            ap[mf][1] = *reinterpret_cast<const short8_t*>(Pw + (mf * 16 + l16) * LD + 32 + 8 * g);
        }
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                short8_t bv = *reinterpret_cast<const short8_t*>(Vs[buf] + (df * 16 + l16) * LDK + ks * 32 + 8 * g);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
                    o[mf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, ap[mf][ks], o[mf][df], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();               // the patch is rewritten in the next tile
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {                  // (all lanes: the shuffles come before the row guard)
        l_run[mf] += __shfl_xor(l_run[mf], 16, 64);
        l_run[mf] += __shfl_xor(l_run[mf], 32, 64);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int i = qb + mf * 16 + l16;
        if (i >= Tn) continue;
        const float inv = l_run[mf] > 0.f ? 1.f / l_run[mf] : 0.f;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            uint2 pk;
            pk.x = pack_bf16x2(o[mf][df][0] * inv, o[mf][df][1] * inv);
            pk.y = pack_bf16x2(o[mf][df][2] * inv, o[mf][df][3] * inv);
            *reinterpret_cast<uint2*>(out + (long)i * ldo + df * 16 + 4 * g) = pk;
        }
    }
}


// ------------------------------------------------------------------------------------------ flash with the ESPnet rel-pos term
// Conformer encoder attention (speech/cosyvoice/transformer/attention.py:215-330, RelPositionMultiHeadedAttention) on the
// MFMA, bf16 build:   score(i, j) = ((q_i + u) . k_j + (q_i + v) . p[T - 1 - i + j]) * scale,   p = linear_pos(pos_emb), 2T - 1 rows
// (rel_shift folded into the index, as attn_dense_kernel does).  Same transposed flash scheme as attn_flash_kernel, 16
// queries per wave: per 64-key tile the positions a wave needs are the 79 rows p[mbase ..], mbase = T - 1 - (qb + 15) + j0,
// so BD^T = P_window (Q + v)^T is five more 16x16x32 MFMA pairs (A = the window rows, read from global two tiles ahead into
// registers, B = the q + v fragment); lane (query l16) then needs BD^T[jj + 15 - l16] for its keys jj - a per-lane shift
// of the row index, done through the wave's LDS patch (float4 writes [query][80 positions], four dword reads per key
// fragment).  The patch is reused for P afterwards.
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_relpos_kernel(
    const bf16_t* __restrict__ q, long ldq, long q_bs, const bf16_t* __restrict__ k, long ldk, long k_bs,
    const bf16_t* __restrict__ vt, long ldvt, long vt_bs, const bf16_t* __restrict__ pos, long ldp,
    const float* __restrict__ pos_u, const float* __restrict__ pos_v, bf16_t* __restrict__ out, long ldo, long o_bs,
    int Tn, float scale, int chunk, int nq, int nheads, int npairs, const int32_t* __restrict__ klen) {
    constexpr int D = 64, KT = 64, LDK = 80, LD = 72, BW = 84;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][KT * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][D * LDK];
    __shared__ __attribute__((aligned(16))) float Bp[NW][16 * BW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd;
    if (pair >= npairs) return;
    const int qt = slot % nq;
    const int b = pair / nheads, h = pair % nheads;
    const int qb = qt * (NW * 16) + wave * 16;
    q += (long)b * q_bs + h * D;
    k += (long)b * k_bs + h * D;
    vt += (long)b * vt_bs + (long)h * D * ldvt;
    out += (long)b * o_bs + h * D;
    pos += h * D;
    const float sc2 = scale * 1.44269504088896341f;
    // klen: valid rows of batch member b in a padded batch (keys beyond it are masked; the relative position of a pair does
    // not depend on the sequence length, so the table of the padded length serves every member)
    const int Tk = klen ? (klen[b] < Tn ? klen[b] : Tn) : Tn;

    short8_t aqu[2], aqv[2];                           // lane (q = l16, k-group g): (q + u), (q + v) [ks*32 + 8g .. +7]
    {
        int row = qb + l16;
        row = row < Tn ? row : Tn - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 raw = *reinterpret_cast<const uint4*>(q + (long)row * ldq + ks * 32 + 8 * g);
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
            const float* pu = pos_u + h * D + ks * 32 + 8 * g;
            const float* pv = pos_v + h * D + ks * 32 + 8 * g;
            unsigned ou[4], ov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = __uint_as_float(w[e] << 16), x1 = __uint_as_float(w[e] & 0xffff0000u);
                ou[e] = pack_bf16x2(x0 + pu[2 * e], x1 + pu[2 * e + 1]);
                ov[e] = pack_bf16x2(x0 + pv[2 * e], x1 + pv[2 * e + 1]);
            }
            aqu[ks] = __builtin_bit_cast(short8_t, make_uint4(ou[0], ou[1], ou[2], ou[3]));
            aqv[ks] = __builtin_bit_cast(short8_t, make_uint4(ov[0], ov[1], ov[2], ov[3]));
        }
    }
    float4_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    int lim = Tk;
    if (chunk > 0) { const int c2 = ((qb + l16) / chunk + 1) * chunk; lim = c2 < lim ? c2 : lim; }
    int kend = Tk;
    if (chunk > 0) {
        int qlast = qt * (NW * 16) + NW * 16 - 1;
        if (qlast > Tn - 1) qlast = Tn - 1;
        const int e = (qlast / chunk + 1) * chunk;
        if (e < kend) kend = e;
    }
    const int ntile = (kend + KT - 1) / KT;
    int vis_all = Tk;
    if (chunk > 0) {
        const int e = ((qt * (NW * 16)) / chunk + 1) * chunk;
        if (e < vis_all) vis_all = e;
    }
    float* Bw = Bp[wave];
    bf16_t* Pw = reinterpret_cast<bf16_t*>(Bw);

    uint4 kreg[512 / (64 * NW)], vreg[512 / (64 * NW)];
    auto load_tiles = [&](int j0) {
#pragma unroll
        for (int i = 0; i < 512 / (64 * NW); ++i) {
            const int id = tid + i * 64 * NW;
            const int r = id >> 3, c = (id & 7) * 8;
            const int key = j0 + r;
            kreg[i] = key < Tk ? *reinterpret_cast<const uint4*>(k + (long)key * ldk + c) : make_uint4(0, 0, 0, 0);
            vreg[i] = (j0 + c < Tk) ? *reinterpret_cast<const uint4*>(vt + (long)r * ldvt + j0 + c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 512 / (64 * NW); ++i) {
            const int id = tid + i * 64 * NW;
            const int r = id >> 3, c = (id & 7) * 8;
            *reinterpret_cast<uint4*>(Ks[buf] + r * LDK + c) = kreg[i];
            *reinterpret_cast<uint4*>(Vs[buf] + r * LDK + c) = vreg[i];
        }
    };
    // the position window of tile j0: rows mbase + f*16 + l16 (clamped: rows outside [0, 2T-2] only meet masked pairs)
    short8_t pw[5][2];
    const int prow_max = 2 * Tn - 2;
    auto load_window = [&](int j0) {
        const int mbase = Tn - 1 - (qb + 15) + j0;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            int row = mbase + f * 16 + l16;
            row = row < 0 ? 0 : (row > prow_max ? prow_max : row);
            const bf16_t* pp = pos + (long)row * ldp + 8 * g;
            pw[f][0] = *reinterpret_cast<const short8_t*>(pp);
            pw[f][1] = *reinterpret_cast<const short8_t*>(pp + 32);
        }
    };
    load_tiles(0);
    store_tiles(0);
    if (ntile > 1) load_tiles(KT);
    load_window(0);
    for (int jt = 0; jt < ntile; ++jt) {
        const int j0 = jt * KT, buf = jt & 1;
        __syncthreads();
        if (jt + 1 < ntile) store_tiles(buf ^ 1);
        if (jt + 2 < ntile) load_tiles(j0 + 2 * KT);
        // BD^T: rows = window positions f*16 + 4g + r, column = query l16
        float4_t bd[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            bd[f] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bd[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[f][ks], aqv[ks], bd[f], 0, 0, 0);
        }
        if (jt + 1 < ntile) load_window(j0 + KT);       // lands under the rest of this tile
#pragma unroll
        for (int f = 0; f < 5; ++f) *reinterpret_cast<float4_t*>(Bw + l16 * BW + f * 16 + 4 * g) = bd[f];
        // AC^T = K (Q + u)^T
        float4_t s[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            s[nf] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const short8_t bk = *reinterpret_cast<const short8_t*>(Ks[buf] + (nf * 16 + l16) * LDK + ks * 32 + 8 * g);
                s[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk, aqu[ks], s[nf], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const float* src = Bw + l16 * BW + 15 - l16 + 4 * g;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[nf][r] += src[nf * 16 + r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();               // every lane has read the window before P overwrites the patch
        const bool need_mask = j0 + KT > vis_all;      // uniform per tile (vis_all <= Tk)
        float mx = -INFINITY;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s[nf][r];
                if (need_mask) {
                    const int j = j0 + nf * 16 + 4 * g + r;
                    x = j < lim ? x : -INFINITY;
                    s[nf][r] = x;
                }
                mx = fmaxf(mx, x);
            }
        mx *= sc2;
        float m_use = m_run;
        const bool grow = (mx - m_run) > 6.0f || m_run == -INFINITY;
        if (__any(grow)) {
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
            l_run *= alpha;
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[df][r] *= alpha;
            m_run = m_new;
            m_use = m_safe;
        }
        float rs = 0.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][0], sc2, -m_use));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][1], sc2, -m_use));
            const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][2], sc2, -m_use));
            const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][3], sc2, -m_use));
            rs += (p0 + p1) + (p2 + p3);
            uint2 pk;
            pk.x = pack_bf16x2(p0, p1);
            pk.y = pack_bf16x2(p2, p3);
            *reinterpret_cast<uint2*>(Pw + l16 * LD + nf * 16 + 4 * g) = pk;
        }
        l_run += rs;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        short8_t ap[2];
        ap[0] = *reinterpret_cast<const short8_t*>(Pw + l16 * LD + 8 * g);
        ap[1] = *reinterpret_cast<const short8_t*>(Pw + l16 * LD + 32 + 8 * g);
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const short8_t bv = *reinterpret_cast<const short8_t*>(Vs[buf] + (df * 16 + l16) * LDK + ks * 32 + 8 * g);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, ap[ks], o[df], 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();               // the patch is rewritten in the next tile
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int i = qb + l16;
    if (i < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            uint2 pk;
            pk.x = pack_bf16x2(o[df][0] * inv, o[df][1] * inv);
            pk.y = pack_bf16x2(o[df][2] * inv, o[df][3] * inv);
            *reinterpret_cast<uint2*>(out + (long)i * ldo + df * 16 + 4 * g) = pk;
        }
    }
}

extern "C" int mmx_attn_relpos_bf16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                                    const void* vt, int64_t ldvt, int64_t vt_bs, const void* pos, int64_t ldp,
                                    const float* pos_u, const float* pos_v, void* out, int64_t ldo, int64_t o_bs,
                                    int B, int H, int T_, float scale, int chunk, const int32_t* klen, hipStream_t stream) {
    MMX_CHECK_ARG(q && k && vt && pos && pos_u && pos_v && out && B > 0 && H > 0 && T_ > 0 && chunk >= 0);
    MMX_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldp % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && vt_bs % 8 == 0);
    MMX_CHECK_ARG(ldvt >= ((T_ + 7) / 8) * 8 && ldo % 4 == 0 && o_bs % 4 == 0);
    MMX_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)pos % 16) == 0 && ((uintptr_t)out % 8) == 0);
    const int npairs = H * B, nq = (T_ + 63) / 64;
    hipLaunchKernelGGL((attn_relpos_kernel<4>), dim3(8 * ((npairs + 7) / 8) * nq), dim3(256), 0, stream, (const bf16_t*)q, ldq, q_bs,
                       (const bf16_t*)k, ldk, k_bs, (const bf16_t*)vt, ldvt, vt_bs, (const bf16_t*)pos, ldp, pos_u, pos_v, (bf16_t*)out, ldo,
                       o_bs, T_, scale, chunk, nq, H, npairs, klen);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}


// ------------------------------------------------------------------------------------------ flash, keys split over waves
// Few queries against many keys (a streaming hop: ~64 new frames x up to 3 000 cached keys per (batch, head) pair): the
// kernel above gives such a launch 32 workgroups that each walk every key tile in turn (47 tiles, 42 us per launch, 30 %
// of the GPU time of a 60 s streaming utterance).  Here a workgroup takes ONE 16-query fragment and its 4 waves take
// key tiles w, w+4, w+8, ... (flash decoding inside the workgroup): each wave stages its own K / V^T tiles in its own
// LDS region (register-staged prefetch of its next tile, no workgroup barrier in the loop), keeps its own running
// (max, sum, O^T), and the four partial results are merged once through LDS.  Same arithmetic per tile as above.
__global__ __launch_bounds__(256) void attn_flash_splitk_kernel(
    const bf16_t* __restrict__ q, long ldq, long q_bs, const bf16_t* __restrict__ k, long ldk, long k_bs,
    const bf16_t* __restrict__ vt, long ldvt, long vt_bs, bf16_t* __restrict__ out, long ldo, long o_bs,
    int Tn, float scale, const float* __restrict__ keymask, long km_bs, int chunk, int nq, int nheads, int npairs,
    int q_begin) {
    constexpr int D = 64, KT = 64, LDK = 80, LD = 72;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    bf16_t* Kw = reinterpret_cast<bf16_t*>(smem_raw) + wave * (2 * KT * LDK + 16 * LD);   // this wave's K tile
    bf16_t* Vw = Kw + KT * LDK;                                                         // ... V^T tile
    bf16_t* Pw = Vw + D * LDK;                                                          // ... P patch
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd;
    if (pair >= npairs) return;
    const int qt = slot % nq;
    const int b = pair / nheads, h = pair % nheads;
    const int qb = q_begin + qt * 16;                  // the workgroup's 16 queries
    q += (long)b * q_bs + h * D;
    k += (long)b * k_bs + h * D;
    vt += (long)b * vt_bs + (long)h * D * ldvt;
    out += (long)b * o_bs + h * D;
    const float* km = keymask ? keymask + (long)b * km_bs : nullptr;
    const float sc2 = scale * 1.44269504088896341f;

    short8_t aq[2];
    {
        int row = qb + l16;
        row = row < Tn ? row : Tn - 1;
        const bf16_t* qp = q + (long)row * ldq + 8 * g;
        aq[0] = *reinterpret_cast<const short8_t*>(qp);
        aq[1] = *reinterpret_cast<const short8_t*>(qp + 32);
    }
    float4_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    int lim = Tn;
    {
        const int i = qb + l16;
        if (chunk > 0) { int c2 = (i / chunk + 1) * chunk; lim = c2 < lim ? c2 : lim; }
    }
    int kend = Tn;
    if (chunk > 0) {
        int qlast = qb + 15;
        if (qlast > Tn - 1) qlast = Tn - 1;
        int e = (qlast / chunk + 1) * chunk;
        if (e < kend) kend = e;
    }
    const int ntile = (kend + KT - 1) / KT;
    int vis_all = Tn;                                  // keys below it are visible to all 16 queries: unmasked tiles
    if (chunk > 0) {
        const int e = (qb / chunk + 1) * chunk;
        if (e < vis_all) vis_all = e;
    }

    uint4 kreg[8], vreg[8];
    auto load_tiles = [&](int j0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = lane + i * 64;              // 512 chunks of 16 B per tile, all on this wave
            const int r = id >> 3, c = (id & 7) * 8;
            const int key = j0 + r;
            kreg[i] = key < Tn ? *reinterpret_cast<const uint4*>(k + (long)key * ldk + c) : make_uint4(0, 0, 0, 0);
            vreg[i] = (j0 + c < Tn) ? *reinterpret_cast<const uint4*>(vt + (long)r * ldvt + j0 + c) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = lane + i * 64;
            const int r = id >> 3, c = (id & 7) * 8;
            *reinterpret_cast<uint4*>(Kw + r * LDK + c) = kreg[i];
            *reinterpret_cast<uint4*>(Vw + r * LDK + c) = vreg[i];
        }
    };
    if (wave < ntile) load_tiles(wave * KT);
    for (int jt = wave; jt < ntile; jt += 4) {
        const int j0 = jt * KT;
        __builtin_amdgcn_wave_barrier();               // every lane is done reading the previous tile
        store_tiles();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (jt + 4 < ntile) load_tiles(j0 + 4 * KT);
        float4_t s[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            s[nf] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const short8_t bk = *reinterpret_cast<const short8_t*>(Kw + (nf * 16 + l16) * LDK + ks * 32 + 8 * g);
                s[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk, aq[ks], s[nf], 0, 0, 0);
            }
        }
        const bool need_mask = km || (j0 + KT > vis_all);
        float mx = -INFINITY;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = s[nf][r];
                if (need_mask) {
                    const int j = j0 + nf * 16 + 4 * g + r;
                    const bool vis = j < lim && j < Tn && (!km || km[j] != 0.f);
                    x = vis ? x : -INFINITY;
                    s[nf][r] = x;
                }
                mx = fmaxf(mx, x);
            }
        mx *= sc2;
        // as in attn_flash_kernel: the running max moves (with one cross-lane maximum) only when some score of the wave
        // is more than 2^6 above it; a steady-state tile has no shuffle
        float m_safe = m_run;
        const bool grow = (mx - m_run) > 6.0f || m_run == -INFINITY;
        if (__any(grow)) {
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            m_safe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
            l_run *= alpha;
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[df][r] *= alpha;
            m_run = m_new;
        }
        float rs = 0.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][0], sc2, -m_safe));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][1], sc2, -m_safe));
            const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][2], sc2, -m_safe));
            const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][3], sc2, -m_safe));
            rs += (p0 + p1) + (p2 + p3);
            uint2 pk;
            pk.x = pack_bf16x2(p0, p1);
            pk.y = pack_bf16x2(p2, p3);
            *reinterpret_cast<uint2*>(Pw + l16 * LD + nf * 16 + 4 * g) = pk;
        }
        l_run += rs;                                   // per-lane partial sum; added up over the query's four lanes below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const short8_t ap0 = *reinterpret_cast<const short8_t*>(Pw + l16 * LD + 8 * g);
        const short8_t ap1 = *reinterpret_cast<const short8_t*>(Pw + l16 * LD + 32 + 8 * g);
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const short8_t bv0 = *reinterpret_cast<const short8_t*>(Vw + (df * 16 + l16) * LDK + 8 * g);
            const short8_t bv1 = *reinterpret_cast<const short8_t*>(Vw + (df * 16 + l16) * LDK + 32 + 8 * g);
            o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv0, ap0, o[df], 0, 0, 0);
            o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv1, ap1, o[df], 0, 0, 0);
        }
    }
    // ---- merge the four waves' partial results: part[w][q][d] fp32, (m, l) per (w, q)
    __syncthreads();                                   // every wave is done with its tiles: the region is reused
    float* part = reinterpret_cast<float*>(smem_raw);  // [4][16][64 + 1]
    float* pm = part + 4 * 16 * 65;                    // [4][16]
    float* pl = pm + 64;                               // [4][16]
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(wave * 16 + l16) * 65 + df * 16 + 4 * g + r] = o[df][r];
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g == 0) { pm[wave * 16 + l16] = m_run; pl[wave * 16 + l16] = l_run; }
    __syncthreads();
    {
        // thread -> (query qi = tid >> 4, 4 channels d0 = (tid & 15) * 4)
        const int qi = tid >> 4, d0 = (tid & 15) * 4;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, pm[w * 16 + qi]);
        float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = pm[w * 16 + qi];
            const float wgt = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - M);
            L += pl[w * 16 + qi] * wgt;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += part[(w * 16 + qi) * 65 + d0 + e] * wgt;
        }
        const int i = qb + qi;
        if (i < Tn) {
            const float inv = L > 0.f ? 1.f / L : 0.f;
            uint2 pk;
            pk.x = pack_bf16x2(acc[0] * inv, acc[1] * inv);
            pk.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
            *reinterpret_cast<uint2*>(out + (long)i * ldo + d0) = pk;
        }
    }
}

extern "C" int mmx_attn_flash_bf16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                                   const void* vt, int64_t ldvt, int64_t vt_bs, void* out, int64_t ldo, int64_t o_bs,
                                   int B, int H, int T_, float scale, const float* keymask, int64_t km_bs, int chunk,
                                   int q_begin, const int32_t* klen, hipStream_t stream) {
    MMX_CHECK_ARG(q && k && vt && out && B > 0 && H > 0 && T_ > 0 && chunk >= 0);
    MMX_CHECK_ARG(q_begin >= 0 && q_begin < T_ && q_begin % 16 == 0);
    MMX_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && vt_bs % 8 == 0);
    MMX_CHECK_ARG(ldvt >= ((T_ + 7) / 8) * 8);
    MMX_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0);
    const int npairs = H * B;
    const int Tq = T_ - q_begin;
    if ((long)npairs * ((Tq + 63) / 64) < 96 && T_ >= 512 && !klen) {            // few queries, many keys: split the keys over waves
        const int nq16 = (Tq + 15) / 16;
        const size_t lds = 4 * (2 * 64 * 80 + 16 * 72) * sizeof(bf16_t);   // >= the merge buffers (4*16*65 + 128 floats)
        MMX_LDS_OPT_IN(attn_flash_splitk_kernel, lds);
        hipLaunchKernelGGL(attn_flash_splitk_kernel, dim3(8 * ((npairs + 7) / 8) * nq16), dim3(256), lds, stream, (const bf16_t*)q, ldq,
                           q_bs, (const bf16_t*)k, ldk, k_bs, (const bf16_t*)vt, ldvt, vt_bs, (bf16_t*)out, ldo, o_bs, T_, scale, keymask,
                           km_bs, chunk, nq16, H, npairs, q_begin);
        MMX_LAUNCH_CHECK();
        return MMX_OK;
    }
    const bool small = (long)npairs * ((Tq + 127) / 128) < 192;         // fewer 128-query tiles than ~3/4 of the CUs
    const int qtile = small ? 64 : 128, nq = (Tq + qtile - 1) / qtile;
    dim3 grid(8 * ((npairs + 7) / 8) * nq);
    if (small)
        hipLaunchKernelGGL((attn_flash_kernel<1, false>), grid, dim3(256), 0, stream, (const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs,
                           (const bf16_t*)vt, ldvt, vt_bs, (bf16_t*)out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    else
        hipLaunchKernelGGL((attn_flash_kernel<1, false, 8>), grid, dim3(512), 0, stream, (const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs,
                           (const bf16_t*)vt, ldvt, vt_bs, (bf16_t*)out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}


extern "C" int mmx_attn_flash_fp8(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                                  const void* vt, int64_t ldvt, int64_t vt_bs, void* out, int64_t ldo, int64_t o_bs,
                                  int B, int H, int T_, float scale, const float* keymask, int64_t km_bs, int chunk,
                                  int q_begin, const int32_t* klen, hipStream_t stream) {
    MMX_CHECK_ARG(q && k && vt && out && B > 0 && H > 0 && T_ > 0 && chunk >= 0);
    MMX_CHECK_ARG(q_begin >= 0 && q_begin < T_ && q_begin % 16 == 0);
    MMX_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && vt_bs % 8 == 0);
    MMX_CHECK_ARG(ldvt >= ((T_ + 7) / 8) * 8);
    MMX_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0);
    const int npairs = H * B, Tq = T_ - q_begin;
    const bool small = (long)npairs * ((Tq + 127) / 128) < 192;
    const int qtile = small ? 64 : 128, nq = (Tq + qtile - 1) / qtile;
    dim3 grid(8 * ((npairs + 7) / 8) * nq);
    if (small)
        hipLaunchKernelGGL((attn_flash_kernel<1, true>), grid, dim3(256), 0, stream, (const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs,
                           (const bf16_t*)vt, ldvt, vt_bs, (bf16_t*)out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    else
        hipLaunchKernelGGL((attn_flash_kernel<2, true>), grid, dim3(256), 0, stream, (const bf16_t*)q, ldq, q_bs, (const bf16_t*)k, ldk, k_bs,
                           (const bf16_t*)vt, ldvt, vt_bs, (bf16_t*)out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
