// Flash attention of the split build (MMX_X2): q / k / v and the output are fp32 in HBM, the products run on the bf16 MFMA
// with BOTH operands of each product split into bf16 hi + lo (both are activations here, unlike the GEMMs, where the
// weights are exact bf16): S = Qh Kh + Ql Kh + Qh Kl and O = Ph Vh + Pl Vh + Ph Vl — three MFMAs per product, the
// lo x lo term (2^-18 relative) is dropped.  Same contract as mmx_attn_flash_bf16 except that V comes ROW-major
// (v[b][t][h*D + d], the QKV projection's own output): the workgroup transposes its V tile on the way into LDS.
//
// Structure (as attn_flash_kernel, csrc/attention.hip): block = 4 waves x 16*MF queries, key tiles of 64, everything
// computed transposed (S^T = K Q^T, O^T = V^T P^T) so a lane owns one query; K and V^T tiles double buffered in LDS with
// register-staged prefetch two tiles ahead, one workgroup barrier per key tile; lazy-rescale online softmax.
// LDS images are UNPADDED 128-byte rows (64 bf16) with the 16-byte chunk index XOR-ed with (row & 7): the ds_read_b128
// fragment reads (16 rows x 2 adjacent chunks per 16-lane service group, MI355X_MICROARCH.md "LDS") then land on 16
// distinct 16-byte bank slots, and the tile needs no pad columns (hi + lo planes of K, V^T and P: 96 KB at MF = 2).
#include "common.h"
#include "../../include/mmx_hip.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

// byte offset of 16-byte chunk `c` (0..7) of 128-byte row `row`
__device__ __forceinline__ int swz(int row, int c) { return row * 128 + ((c ^ (row & 7)) << 4); }

__device__ __forceinline__ void split4(const float x[4], uint2& hi, uint2& lo) {
    hi.x = pack_bf16x2(x[0], x[1]);
    hi.y = pack_bf16x2(x[2], x[3]);
    lo.x = pack_bf16x2(x[0] - __uint_as_float(hi.x << 16), x[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(x[2] - __uint_as_float(hi.y << 16), x[3] - __uint_as_float(hi.y & 0xffff0000u));
}

// PRE = false: q / k / v fp32 row-major, split (and V transposed) by this kernel tile by tile.
// PRE = true : the PRODUCER already split them (csrc/fused.hip ln_qkv, split build): q / k point at bf16 rows [hi Q | hi K |
//              lo Q | lo K] (the lo terms 1024 columns behind the hi terms), v at V TRANSPOSED as bf16 planes
//              vt[plane][h*D + d][t] (plane stride 512 * ldv); tiles go from HBM to LDS as they are.  Every key / value tile
//              is used by all query tiles of its (batch, head): splitting it once at the producer instead of once per query
//              tile removes the conversion VALU work (and its shuffles) from this kernel's loop.
// NW waves of 16 * MF queries each.  NW = 8, MF = 1 (two waves per SIMD): with three MFMAs per product and the hi / lo split
// of P, one wave per SIMD runs its MFMA phase and its softmax / split VALU phase one after the other (~1500 cycles each per
// key tile); two waves per SIMD overlap one's VALU with the other's MFMAs on the same SIMD for the same LDS footprint.
template <int MF, bool PRE, int NW>
__global__ __launch_bounds__(64 * NW) void attn_flash_x_kernel(
    const void* __restrict__ q_, long ldq, long q_bs, const void* __restrict__ k_, long ldk, long k_bs,
    const void* __restrict__ v_, long ldv, long v_bs, float* __restrict__ out, long ldo, long o_bs,
    int Tn, float scale, const float* __restrict__ keymask, long km_bs, int chunk, int nq, int nheads, int npairs,
    int q_begin, const int32_t* __restrict__ klen) {
    typedef std::conditional_t<PRE, bf16_t, float> TI;
    const TI* q = reinterpret_cast<const TI*>(q_);
    const TI* k = reinterpret_cast<const TI*>(k_);
    const TI* v = reinterpret_cast<const TI*>(v_);
    constexpr int D = 64, KT = 64, QW = 16 * MF;
    constexpr int TILE = KT * 128;                     // bytes of one [64][64] bf16 image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kh = smem;                                   // [2 bufs][TILE]   keys x channels
    char* Kl = Kh + 2 * TILE;
    char* Vh = Kl + 2 * TILE;                          // [2 bufs][TILE]   channels x keys (V^T)
    char* Vl = Vh + 2 * TILE;
    char* Ph = Vl + 2 * TILE;                          // [4 waves][QW rows x 128 B]   queries x keys
    char* Pl = Ph + NW * QW * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd;            // XCD-aware: the query tiles of one (batch, head) share an L2
    if (pair >= npairs) return;
    const int qt = slot % nq;
    const int b = pair / nheads, h = pair % nheads;
    const int qb = q_begin + qt * (NW * QW) + wave * QW;
    q += (long)b * q_bs + h * D;
    k += (long)b * k_bs + h * D;
    v += PRE ? (long)b * v_bs + (long)h * D * ldv : (long)b * v_bs + h * D;
    out += (long)b * o_bs + h * D;
    const float* km = keymask ? keymask + (long)b * km_bs : nullptr;
    const float sc2 = scale * 1.44269504088896341f;
    const int Tk = klen ? (klen[b] < Tn ? klen[b] : Tn) : Tn;
    if (klen && q_begin + qt * (NW * QW) >= Tk) {        // a workgroup of pure padding rows
        for (int id = tid; id < NW * QW * 16; id += 64 * NW) {
            const int i = q_begin + qt * (NW * QW) + (id >> 4);
            if (i < Tn) *reinterpret_cast<float4*>(out + (long)i * ldo + (id & 15) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    // Q fragments, hi + lo: lane (query l16, k-group g) holds Q[q][ks*32 + 8g .. +7]
    short8_t aqh[MF][2], aql[MF][2];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int row = qb + mf * 16 + l16;
        row = row < Tn ? row : Tn - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (PRE) {
                const bf16_t* qp = reinterpret_cast<const bf16_t*>(q) + (long)row * ldq + ks * 32 + 8 * g;
                aqh[mf][ks] = *reinterpret_cast<const short8_t*>(qp);
                aql[mf][ks] = *reinterpret_cast<const short8_t*>(qp + 1024);
            } else {
                const float* qp = reinterpret_cast<const float*>(q) + (long)row * ldq + ks * 32 + 8 * g;
                const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
                const float x0[4] = {a.x, a.y, a.z, a.w}, x1[4] = {c.x, c.y, c.z, c.w};
                uint2 h0, l0, h1, l1;
                split4(x0, h0, l0);
                split4(x1, h1, l1);
                aqh[mf][ks] = __builtin_bit_cast(short8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
                aql[mf][ks] = __builtin_bit_cast(short8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
            }
        }
    }
    float4_t o[MF][4];
    float m_run[MF], l_run[MF];
    int lim[MF];
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
    int kend = Tk;
    if (chunk > 0) {
        int qlast = q_begin + qt * (NW * QW) + NW * QW - 1;
        if (qlast > Tn - 1) qlast = Tn - 1;
        int e = (qlast / chunk + 1) * chunk;
        if (e < kend) kend = e;
    }
    const int ntile = (kend + KT - 1) / KT;
    int vis_all = Tk;
    if (chunk > 0) {
        const int e = ((q_begin + qt * (NW * QW)) / chunk + 1) * chunk;
        if (e < vis_all) vis_all = e;
    }
    char* Pwh = Ph + wave * QW * 128;
    char* Pwl = Pl + wave * QW * 128;

    // tile loads: 1024 chunks of 4 floats per operand, 4 per thread.
    //   K: chunk id -> (key r = id >> 4, channels 4*(id & 15) ..): coalesced 256-byte rows
    //   V: chunk id -> (key 2*(id >> 5) + (id & 1), channels 4*((id >> 1) & 15) ..): ADJACENT LANES hold the two keys of a
    //      pair for the same channels; they swap halves (one shuffle pair) so each lane owns 2 channels x 2 keys and writes
    //      (key 2p, key 2p+1) as one dword of the transposed image
    constexpr int NT = 64 * NW, CPT = 1024 / NT;       // threads; fp32 chunks (or 2 x bf16 chunks) per thread, tile and operand
    float4 kreg[CPT], vreg[CPT];
    auto load_tiles = [&](int j0) {
        if constexpr (PRE) {
            // 512 chunks of 8 bf16 per plane and operand: chunk id -> (row r = id >> 3, 16-byte chunk id & 7); kreg / vreg
            // [2p + plane]: p-th chunk of this thread, hi / lo plane
            const bf16_t* kb = reinterpret_cast<const bf16_t*>(k);
            const bf16_t* vb = reinterpret_cast<const bf16_t*>(v);
#pragma unroll
            for (int pi = 0; pi < CPT / 2; ++pi) {
                const int id = tid + pi * NT, r = id >> 3, c = (id & 7) * 8;
                const int key = j0 + r;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    kreg[2 * pi + pl] = key < Tk ? *reinterpret_cast<const float4*>(kb + (long)key * ldk + pl * 1024 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    // V^T rows are channels; columns j0 + c .. +7 (the buffer is zero padded to a multiple of 8 columns)
                    vreg[2 * pi + pl] = (j0 + c < Tk) ? *reinterpret_cast<const float4*>(vb + (long)pl * 512 * ldv + (long)r * ldv + j0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int id = tid + i * NT;
            const int kr = j0 + (id >> 4);
            kreg[i] = kr < Tk ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(k) + (long)kr * ldk + (id & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int vr = j0 + 2 * (id >> 5) + (id & 1);
            vreg[i] = vr < Tk ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(v) + (long)vr * ldv + ((id >> 1) & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&](int buf) {
        if constexpr (PRE) {
#pragma unroll
            for (int pi = 0; pi < CPT / 2; ++pi) {
                const int id = tid + pi * NT, r = id >> 3, c8 = id & 7;
                const int off = buf * TILE + swz(r, c8);
                *reinterpret_cast<float4*>(Kh + off) = kreg[2 * pi];
                *reinterpret_cast<float4*>(Kl + off) = kreg[2 * pi + 1];
                *reinterpret_cast<float4*>(Vh + off) = vreg[2 * pi];
                *reinterpret_cast<float4*>(Vl + off) = vreg[2 * pi + 1];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int id = tid + i * NT;
            {
                const int r = id >> 4, c4 = id & 15;   // 4 channels = half a 16-byte chunk of bf16
                const float x[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
                uint2 hi, lo;
                split4(x, hi, lo);
                const int off = buf * TILE + swz(r, c4 >> 1) + (c4 & 1) * 8;
                *reinterpret_cast<uint2*>(Kh + off) = hi;
                *reinterpret_cast<uint2*>(Kl + off) = lo;
            }
            {
                const int par = id & 1, cc = (id >> 1) & 15, kp = id >> 5;
                // even lane keeps channels 4cc, 4cc+1 and receives them of key 2kp+1; odd lane keeps 4cc+2, 4cc+3
                const float s0 = par ? vreg[i].x : vreg[i].z, s1 = par ? vreg[i].y : vreg[i].w;
                const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
                const float a0 = par ? r0 : vreg[i].x, b0 = par ? vreg[i].z : r0;     // channel d0: (key 2kp, key 2kp+1)
                const float a1 = par ? r1 : vreg[i].y, b1 = par ? vreg[i].w : r1;     // channel d0 + 1
                const int d0 = 4 * cc + 2 * par;
                const unsigned h0 = pack_bf16x2(a0, b0), h1 = pack_bf16x2(a1, b1);
                const unsigned l0 = pack_bf16x2(a0 - __uint_as_float(h0 << 16), b0 - __uint_as_float(h0 & 0xffff0000u));
                const unsigned l1 = pack_bf16x2(a1 - __uint_as_float(h1 << 16), b1 - __uint_as_float(h1 & 0xffff0000u));
                const int o0 = buf * TILE + swz(d0, kp >> 2) + (kp & 3) * 4, o1 = buf * TILE + swz(d0 + 1, kp >> 2) + (kp & 3) * 4;
                *reinterpret_cast<unsigned*>(Vh + o0) = h0;
                *reinterpret_cast<unsigned*>(Vl + o0) = l0;
                *reinterpret_cast<unsigned*>(Vh + o1) = h1;
                *reinterpret_cast<unsigned*>(Vl + o1) = l1;
            }
        }
    };
    load_tiles(0);
    store_tiles(0);
    if (ntile > 1) load_tiles(KT);
    for (int jt = 0; jt < ntile; ++jt) {
        const int j0 = jt * KT, buf = jt & 1;
        __syncthreads();
        if (jt + 1 < ntile) store_tiles(buf ^ 1);
        if (jt + 2 < ntile) load_tiles(j0 + 2 * KT);
        // S^T = K Q^T
        float4_t s[MF][4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) s[mf][nf] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = buf * TILE + swz(nf * 16 + l16, ks * 4 + g);
                const short8_t bkh = *reinterpret_cast<const short8_t*>(Kh + off);
                const short8_t bkl = *reinterpret_cast<const short8_t*>(Kl + off);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    s[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkh, aqh[mf][ks], s[mf][nf], 0, 0, 0);
                    s[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkh, aql[mf][ks], s[mf][nf], 0, 0, 0);
                    s[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkl, aqh[mf][ks], s[mf][nf], 0, 0, 0);
                }
            }
        }
        const bool need_mask = km || (j0 + KT > vis_all);
        bool kvis[4][4];
        if (need_mask) {
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + nf * 16 + 4 * g + r;
                    kvis[nf][r] = j < Tk && (!km || km[j] != 0.f);
                }
        }
        auto softmax_tile = [&](auto mask_c, auto mf_c) {
            constexpr bool MASK = decltype(mask_c)::value;
            constexpr int mf = decltype(mf_c)::value;
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
            mx *= sc2;
            float m_use = m_run[mf];
            const bool grow = (mx - m_run[mf]) > 6.0f || m_run[mf] == -INFINITY;     // lazy rescale (see attention.hip)
            if (__any(grow)) {
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[mf], mx);
                const float m_safe = m_new == -INFINITY ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[mf] - m_safe);
                l_run[mf] *= alpha;
#pragma unroll
                for (int df = 0; df < 4; ++df)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[mf][df][r] *= alpha;
                m_run[mf] = m_new;
                m_use = m_safe;
            }
            float rs = 0.f;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mf][nf][r], sc2, -m_use));
                rs += (p[0] + p[1]) + (p[2] + p[3]);
                uint2 hi, lo;
                split4(p, hi, lo);
                // keys nf*16 + 4g .. +3 of query row mf*16 + l16: half a chunk
                const int off = swz(mf * 16 + l16, nf * 2 + (g >> 1)) + (g & 1) * 8;
                *reinterpret_cast<uint2*>(Pwh + off) = hi;
                *reinterpret_cast<uint2*>(Pwl + off) = lo;
            }
            l_run[mf] += rs;
        };
        if (need_mask) {
            softmax_tile(std::true_type{}, std::integral_constant<int, 0>{});
            if constexpr (MF > 1) softmax_tile(std::true_type{}, std::integral_constant<int, MF - 1>{});
        } else {
            softmax_tile(std::false_type{}, std::integral_constant<int, 0>{});
            if constexpr (MF > 1) softmax_tile(std::false_type{}, std::integral_constant<int, MF - 1>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        short8_t aph[MF][2], apl[MF][2];               // lane (q = l16, g): P[q][ks*32 + 8g .. +7]
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = swz(mf * 16 + l16, ks * 4 + g);
                aph[mf][ks] = *reinterpret_cast<const short8_t*>(Pwh + off);
                apl[mf][ks] = *reinterpret_cast<const short8_t*>(Pwl + off);
            }
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = buf * TILE + swz(df * 16 + l16, ks * 4 + g);
                const short8_t bvh = *reinterpret_cast<const short8_t*>(Vh + off);
                const short8_t bvl = *reinterpret_cast<const short8_t*>(Vl + off);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    o[mf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvh, aph[mf][ks], o[mf][df], 0, 0, 0);
                    o[mf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvh, apl[mf][ks], o[mf][df], 0, 0, 0);
                    o[mf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvl, aph[mf][ks], o[mf][df], 0, 0, 0);
                }
            }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        l_run[mf] += __shfl_xor(l_run[mf], 16, 64);
        l_run[mf] += __shfl_xor(l_run[mf], 32, 64);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int i = qb + mf * 16 + l16;
        if (i >= Tn) continue;
        const float inv = l_run[mf] > 0.f ? 1.f / l_run[mf] : 0.f;
#pragma unroll
        for (int df = 0; df < 4; ++df)
            *reinterpret_cast<float4*>(out + (long)i * ldo + df * 16 + 4 * g) =
                make_float4(o[mf][df][0] * inv, o[mf][df][1] * inv, o[mf][df][2] * inv, o[mf][df][3] * inv);
    }
}

// ------------------------------------------------------------------------------------------ rel-pos attention, split build
// Conformer encoder attention (speech/cosyvoice/transformer/attention.py:215-330, RelPositionMultiHeadedAttention) of the split
// build on the MFMA:   score(i, j) = ((q_i + u) . k_j + (q_i + v) . p[T - 1 - i + j]) * scale,   p = linear_pos(pos_emb), 2T - 1
// rows (rel_shift folded into the index).  attn_flash_x_kernel<1, false, 4> (fp32 q / k / v rows, split into bf16 hi + lo tile by
// tile, three MFMAs per product) plus the position term of attn_relpos_kernel (csrc/attention.hip): per 64-key tile the wave's
// 79-row window of p (fp32 rows, split like every other operand) times (q + v), five 16x16x32 fragment pairs x 3 MFMAs, shifted
// per lane through the wave's fp32 LDS patch.  Replaces attn_dense_kernel<float> (fp32 VALU, 416 us per batched launch).
__global__ __launch_bounds__(256) void attn_relpos_x_kernel(
    const float* __restrict__ q, long ldq, long q_bs, const float* __restrict__ k, long ldk, long k_bs,
    const float* __restrict__ v, long ldv, long v_bs, const float* __restrict__ pos, long ldp,
    const float* __restrict__ pos_u, const float* __restrict__ pos_v, float* __restrict__ out, long ldo, long o_bs,
    int Tn, float scale, int chunk, int nq, int nheads, int npairs, const int32_t* __restrict__ klen) {
    constexpr int D = 64, KT = 64, NW = 4, QW = 16, BW = 84;
    constexpr int TILE = KT * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kh = smem;
    char* Kl = Kh + 2 * TILE;
    char* Vh = Kl + 2 * TILE;
    char* Vl = Vh + 2 * TILE;
    char* Ph = Vl + 2 * TILE;                          // [4 waves][16 rows x 128 B]
    char* Pl = Ph + NW * QW * 128;
    float* Bp = reinterpret_cast<float*>(Pl + NW * QW * 128);   // [4 waves][16][BW] position-term patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nq) * 8 + xcd;
    if (pair >= npairs) return;
    const int qt = slot % nq;
    const int b = pair / nheads, h = pair % nheads;
    const int qb = qt * (NW * QW) + wave * QW;
    q += (long)b * q_bs + h * D;
    k += (long)b * k_bs + h * D;
    v += (long)b * v_bs + h * D;
    out += (long)b * o_bs + h * D;
    pos += h * D;
    const float sc2 = scale * 1.44269504088896341f;
    const int Tk = klen ? (klen[b] < Tn ? klen[b] : Tn) : Tn;

    // (q + u) and (q + v) fragments, hi + lo: lane (query l16, k-group g) holds [ks*32 + 8g .. +7]
    short8_t auh[2], aul[2], avh[2], avl[2];
    {
        int row = qb + l16;
        row = row < Tn ? row : Tn - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* qp = q + (long)row * ldq + ks * 32 + 8 * g;
            const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
            const float* pu = pos_u + h * D + ks * 32 + 8 * g;
            const float* pv = pos_v + h * D + ks * 32 + 8 * g;
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            float xu0[4], xu1[4], xv0[4], xv1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xu0[e] = x[e] + pu[e]; xu1[e] = x[4 + e] + pu[4 + e];
                xv0[e] = x[e] + pv[e]; xv1[e] = x[4 + e] + pv[4 + e];
            }
            uint2 h0, l0, h1, l1;
            split4(xu0, h0, l0);
            split4(xu1, h1, l1);
            auh[ks] = __builtin_bit_cast(short8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
            aul[ks] = __builtin_bit_cast(short8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
            split4(xv0, h0, l0);
            split4(xv1, h1, l1);
            avh[ks] = __builtin_bit_cast(short8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
            avl[ks] = __builtin_bit_cast(short8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
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
        int qlast = qt * (NW * QW) + NW * QW - 1;
        if (qlast > Tn - 1) qlast = Tn - 1;
        const int e = (qlast / chunk + 1) * chunk;
        if (e < kend) kend = e;
    }
    const int ntile = (kend + KT - 1) / KT;
    int vis_all = Tk;
    if (chunk > 0) {
        const int e = ((qt * (NW * QW)) / chunk + 1) * chunk;
        if (e < vis_all) vis_all = e;
    }
    char* Pwh = Ph + wave * QW * 128;
    char* Pwl = Pl + wave * QW * 128;
    float* Bw = Bp + wave * 16 * BW;

    constexpr int NT = 64 * NW, CPT = 1024 / NT;
    float4 kreg[CPT], vreg[CPT];
    auto load_tiles = [&](int j0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int id = tid + i * NT;
            const int kr = j0 + (id >> 4);
            kreg[i] = kr < Tk ? *reinterpret_cast<const float4*>(k + (long)kr * ldk + (id & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int vr = j0 + 2 * (id >> 5) + (id & 1);
            vreg[i] = vr < Tk ? *reinterpret_cast<const float4*>(v + (long)vr * ldv + ((id >> 1) & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&](int buf) {                  // as attn_flash_x_kernel<.., PRE = false, ..>: split K, transpose + split V
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int id = tid + i * NT;
            {
                const int r = id >> 4, c4 = id & 15;
                const float x[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
                uint2 hi, lo;
                split4(x, hi, lo);
                const int off = buf * TILE + swz(r, c4 >> 1) + (c4 & 1) * 8;
                *reinterpret_cast<uint2*>(Kh + off) = hi;
                *reinterpret_cast<uint2*>(Kl + off) = lo;
            }
            {
                const int par = id & 1, cc = (id >> 1) & 15, kp = id >> 5;
                const float s0 = par ? vreg[i].x : vreg[i].z, s1 = par ? vreg[i].y : vreg[i].w;
                const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
                const float a0 = par ? r0 : vreg[i].x, b0 = par ? vreg[i].z : r0;
                const float a1 = par ? r1 : vreg[i].y, b1 = par ? vreg[i].w : r1;
                const int d0 = 4 * cc + 2 * par;
                const unsigned h0 = pack_bf16x2(a0, b0), h1 = pack_bf16x2(a1, b1);
                const unsigned l0 = pack_bf16x2(a0 - __uint_as_float(h0 << 16), b0 - __uint_as_float(h0 & 0xffff0000u));
                const unsigned l1 = pack_bf16x2(a1 - __uint_as_float(h1 << 16), b1 - __uint_as_float(h1 & 0xffff0000u));
                const int o0 = buf * TILE + swz(d0, kp >> 2) + (kp & 3) * 4, o1 = buf * TILE + swz(d0 + 1, kp >> 2) + (kp & 3) * 4;
                *reinterpret_cast<unsigned*>(Vh + o0) = h0;
                *reinterpret_cast<unsigned*>(Vl + o0) = l0;
                *reinterpret_cast<unsigned*>(Vh + o1) = h1;
                *reinterpret_cast<unsigned*>(Vl + o1) = l1;
            }
        }
    };
    // the position window of tile j0: rows mbase + f*16 + l16 (clamped: rows outside [0, 2T-2] only meet masked pairs), fp32, read
    // one tile ahead; split into hi + lo at the top of the tile that uses it
    float4 praw[5][2][2];
    const int prow_max = 2 * Tn - 2;
    auto load_window = [&](int j0) {
        const int mbase = Tn - 1 - (qb + 15) + j0;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            int row = mbase + f * 16 + l16;
            row = row < 0 ? 0 : (row > prow_max ? prow_max : row);
            const float* pp = pos + (long)row * ldp + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                praw[f][ks][0] = *reinterpret_cast<const float4*>(pp + ks * 32);
                praw[f][ks][1] = *reinterpret_cast<const float4*>(pp + ks * 32 + 4);
            }
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
        // BD^T = P_window (Q + v)^T: rows = window positions f*16 + 4g + r, column = query l16
        float4_t bd[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            bd[f] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float x0[4] = {praw[f][ks][0].x, praw[f][ks][0].y, praw[f][ks][0].z, praw[f][ks][0].w};
                const float x1[4] = {praw[f][ks][1].x, praw[f][ks][1].y, praw[f][ks][1].z, praw[f][ks][1].w};
                uint2 h0, l0, h1, l1;
                split4(x0, h0, l0);
                split4(x1, h1, l1);
                const short8_t ph = __builtin_bit_cast(short8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
                const short8_t pl = __builtin_bit_cast(short8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
                bd[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, avh[ks], bd[f], 0, 0, 0);
                bd[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, avl[ks], bd[f], 0, 0, 0);
                bd[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, avh[ks], bd[f], 0, 0, 0);
            }
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
                const int off = buf * TILE + swz(nf * 16 + l16, ks * 4 + g);
                const short8_t bkh = *reinterpret_cast<const short8_t*>(Kh + off);
                const short8_t bkl = *reinterpret_cast<const short8_t*>(Kl + off);
                s[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkh, auh[ks], s[nf], 0, 0, 0);
                s[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkh, aul[ks], s[nf], 0, 0, 0);
                s[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bkl, auh[ks], s[nf], 0, 0, 0);
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
        __builtin_amdgcn_wave_barrier();               // every lane has read the window before the next tile rewrites the patch
        const bool need_mask = j0 + KT > vis_all;
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
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nf][r], sc2, -m_use));
            rs += (p[0] + p[1]) + (p[2] + p[3]);
            uint2 hi, lo;
            split4(p, hi, lo);
            const int off = swz(l16, nf * 2 + (g >> 1)) + (g & 1) * 8;
            *reinterpret_cast<uint2*>(Pwh + off) = hi;
            *reinterpret_cast<uint2*>(Pwl + off) = lo;
        }
        l_run += rs;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        short8_t aph[2], apl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = swz(l16, ks * 4 + g);
            aph[ks] = *reinterpret_cast<const short8_t*>(Pwh + off);
            apl[ks] = *reinterpret_cast<const short8_t*>(Pwl + off);
        }
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = buf * TILE + swz(df * 16 + l16, ks * 4 + g);
                const short8_t bvh = *reinterpret_cast<const short8_t*>(Vh + off);
                const short8_t bvl = *reinterpret_cast<const short8_t*>(Vl + off);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvh, aph[ks], o[df], 0, 0, 0);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvh, apl[ks], o[df], 0, 0, 0);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bvl, aph[ks], o[df], 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int i = qb + l16;
    if (i < Tn) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
        for (int df = 0; df < 4; ++df)
            *reinterpret_cast<float4*>(out + (long)i * ldo + df * 16 + 4 * g) = make_float4(o[df][0] * inv, o[df][1] * inv, o[df][2] * inv, o[df][3] * inv);
    }
}

template <int MF, bool PRE, int NW>
int launch_flash_x(dim3 grid, hipStream_t stream, const void* q, long ldq, long q_bs, const void* k, long ldk, long k_bs,
                   const void* v, long ldv, long v_bs, float* out, long ldo, long o_bs, int T_, float scale,
                   const float* keymask, long km_bs, int chunk, int nq, int H, int npairs, int q_begin, const int32_t* klen) {
    const size_t lds = (size_t)8 * 64 * 128 + (size_t)2 * NW * 16 * MF * 128;
    MMX_LDS_OPT_IN((attn_flash_x_kernel<MF, PRE, NW>), lds);
    hipLaunchKernelGGL((attn_flash_x_kernel<MF, PRE, NW>), grid, dim3(64 * NW), lds, stream, q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs, out, ldo, o_bs, T_,
                       scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

}  // namespace

extern "C" int mmx_attn_flash_x(const float* q, int64_t ldq, int64_t q_bs, const float* k, int64_t ldk, int64_t k_bs,
                                const float* v, int64_t ldv, int64_t v_bs, float* out, int64_t ldo, int64_t o_bs,
                                int B, int H, int T_, float scale, const float* keymask, int64_t km_bs, int chunk,
                                int q_begin, const int32_t* klen, hipStream_t stream) {
    MMX_CHECK_ARG(q && k && v && out && B > 0 && H > 0 && T_ > 0 && chunk >= 0);
    MMX_CHECK_ARG(q_begin >= 0 && q_begin < T_ && q_begin % 16 == 0);
    MMX_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 && v_bs % 4 == 0 && o_bs % 4 == 0);
    MMX_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const int npairs = H * B, Tq = T_ - q_begin;
    const bool small = (long)npairs * ((Tq + 127) / 128) < 192;         // fewer 128-query tiles than ~3/4 of the CUs
    const int qtile = small ? 64 : 128, nq = (Tq + qtile - 1) / qtile;
    dim3 grid(8 * ((npairs + 7) / 8) * nq);
    if (small) return launch_flash_x<1, false, 4>(grid, stream, q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs, out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    return launch_flash_x<1, false, 8>(grid, stream, q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs, out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
}

// The same attention on operands the producer has already split (csrc/fused.hip, split build): qk bf16 [B][T][ldqk >= 2048] =
// [hi Q | hi K | lo Q | lo K], vt bf16 [B][2 planes][512][ldvt] (V transposed, zero padded columns), out fp32.
extern "C" int mmx_attn_flash_xs(const void* qk, int64_t ldqk, int64_t qk_bs, const void* vt, int64_t ldvt, int64_t vt_bs,
                                 float* out, int64_t ldo, int64_t o_bs, int B, int H, int T_, float scale, const float* keymask,
                                 int64_t km_bs, int chunk, int q_begin, const int32_t* klen, int form, hipStream_t stream) {
    MMX_CHECK_ARG(qk && vt && out && B > 0 && H > 0 && H * 64 <= 512 && T_ > 0 && chunk >= 0 && form >= 0 && form <= 3);
    MMX_CHECK_ARG(q_begin >= 0 && q_begin < T_ && q_begin % 16 == 0);
    MMX_CHECK_ARG(ldqk >= 2048 && ldqk % 8 == 0 && qk_bs % 8 == 0 && ldvt % 8 == 0 && ldvt >= ((T_ + 7) / 8) * 8 && vt_bs % 8 == 0 && vt_bs >= 2 * 512 * ldvt);
    MMX_CHECK_ARG(ldo % 4 == 0 && o_bs % 4 == 0 && ((uintptr_t)qk % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 16) == 0);
    const int npairs = H * B, Tq = T_ - q_begin;
    const bool small = form == 3 || (long)npairs * ((Tq + 127) / 128) < 192;
    const int qtile = small ? 64 : 128, nq = (Tq + qtile - 1) / qtile;
    dim3 grid(8 * ((npairs + 7) / 8) * nq);
    const bf16_t* q = (const bf16_t*)qk;
    // 8 waves x 32 queries (256 per workgroup, MF = 2) halve the K / V fragment reads from LDS per MFMA, and a workgroup takes
    // ~1.6 x as long as one of 128 queries (tools/flash_lab.py, profiles/r04_flash_lab_x.txt: 71 -> 62 us at 3 x 980 frames,
    // 37 -> 30 at 5 x 420, 122 -> 112 at 8 x 896; 93 -> 100 at 5 x 860, where the 128-query grid needs 3 rounds of the 256 CUs
    // and the 256-query grid 2).  form 0: chosen per launch from the rounds each grid needs.
    // form 1 (the caller's launch runs BESIDE a latency-bound kernel chain on another stream): the 128-query workgroups - 96 KB of
    // LDS and 136 registers per wave (two waves per SIMD leave 224) instead of 128 KB and 216 (80): a decode workgroup of csrc/decode.hip
    // fits on the same CU.  Measured in the 32-utterance step (gpurun_out/r4_17): decode loop done at 524 ms against 541 (chosen
    // per launch) and 549 (256-query everywhere); 520.8 / 512.4 / 510.7 audio-s/s.  form 2: 256-query workgroups wherever the
    // grid is not small; form 3: the 4-wave 64-query form everywhere (80 KB of LDS).
    const int nq2 = (Tq + 255) / 256;
    const long wg1 = (long)npairs * nq, wg2 = (long)npairs * nq2;
    const bool mf2 = !small && form != 1 && (form == 2 || 1.6 * (double)((wg2 + 255) / 256) < (double)((wg1 + 255) / 256));
    if (mf2) {
        dim3 grid2(8 * ((npairs + 7) / 8) * nq2);
        return launch_flash_x<2, true, 8>(grid2, stream, q, ldqk, qk_bs, q + 512, ldqk, qk_bs, vt, ldvt, vt_bs, out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq2, H, npairs, q_begin, klen);
    }
    // (4 waves x 32 queries - the 256-query form's fragment reuse on 96 KB of LDS, one wave per SIMD - measured 118 us at 5 x 860
    // frames against 92: one wave per SIMD has nothing to overlap its softmax with)
    if (small) return launch_flash_x<1, true, 4>(grid, stream, q, ldqk, qk_bs, q + 512, ldqk, qk_bs, vt, ldvt, vt_bs, out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
    return launch_flash_x<1, true, 8>(grid, stream, q, ldqk, qk_bs, q + 512, ldqk, qk_bs, vt, ldvt, vt_bs, out, ldo, o_bs, T_, scale, keymask, km_bs, chunk, nq, H, npairs, q_begin, klen);
}

// Conformer rel-pos attention of the split build: q / k / v fp32 rows (one [.., 1536] Q | K | V projection), pos fp32 [2T - 1][ldp]
// (linear_pos of the ESPnet table), pos_u / pos_v fp32 [H * 64], out fp32; klen as mmx_attn_flash_x.
// Replaces transformer/attention.py:249-330 (RelPositionMultiHeadedAttention.forward incl. rel_shift :225-247).
extern "C" int mmx_attn_relpos_x(const float* q, int64_t ldq, int64_t q_bs, const float* k, int64_t ldk, int64_t k_bs,
                                 const float* v, int64_t ldv, int64_t v_bs, const float* pos, int64_t ldp, const float* pos_u,
                                 const float* pos_v, float* out, int64_t ldo, int64_t o_bs, int B, int H, int T_, float scale,
                                 int chunk, const int32_t* klen, hipStream_t stream) {
    MMX_CHECK_ARG(q && k && v && pos && pos_u && pos_v && out && B > 0 && H > 0 && T_ > 0 && chunk >= 0);
    MMX_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldp % 4 == 0 && ldo % 4 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 && v_bs % 4 == 0 && o_bs % 4 == 0);
    MMX_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)pos % 16) == 0 &&
                  ((uintptr_t)out % 16) == 0 && ((uintptr_t)pos_u % 16) == 0 && ((uintptr_t)pos_v % 16) == 0);
    const int npairs = H * B, nq = (T_ + 63) / 64;
    const size_t lds = (size_t)8 * 64 * 128 + (size_t)2 * 4 * 16 * 128 + (size_t)4 * 16 * 84 * 4;
    MMX_LDS_OPT_IN(attn_relpos_x_kernel, lds);
    hipLaunchKernelGGL(attn_relpos_x_kernel, dim3(8 * ((npairs + 7) / 8) * nq), dim3(256), lds, stream, q, ldq, q_bs, k, ldk, k_bs, v, ldv, v_bs,
                       pos, ldp, pos_u, pos_v, out, ldo, o_bs, T_, scale, chunk, nq, H, npairs, klen);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
