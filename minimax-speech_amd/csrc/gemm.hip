// Windowed MFMA GEMM for gfx950: the one kernel family behind every Linear, Conv1d and
// ConvTranspose1d of the flow decoder and the DAC-VAE decoder.
//
//   C[b][m][n] = sum_{tap < ntaps} sum_{c < cin}  A[b][(m + tap*dil + row_off)][c] * W[n][tap*cin + c]
//
// A is a time-major activation matrix (row = frame, lda elements per row); a k-tap convolution is a
// GEMM whose A rows are overlapping windows of it, so Conv1d needs no im2col buffer and no physical
// zero padding (rows outside [row_lo,row_hi) read as 0).  ConvTranspose1d (kernel 2s, stride s) is
// the same GEMM with ntaps = 2 and N = s*Cout (all s output phases side by side), whose row-major
// output IS the interleaved signal shifted by `out_off` (see dac-vae/model.py:260-267 for the op).
//
// Tiling: 256 threads = 4 waves (WM x WN); each wave owns (BM/WM) x (BN/WN) of the block tile as
// 16x16 MFMA fragments; K is walked in steps of 32 through a 2-stage LDS ring (register-staged
// prefetch of tile k+1 while tile k is multiplied).  bf16 uses v_mfma_f32_16x16x32_bf16; the fp32
// parity build uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, cdna_hip_programming.md §3).
// LDS rows are padded to a conflict-free pitch (see Frag below).
//
// Fused epilogue: + bias -> activation -> + residual -> * row mask -> store fp32 (residual stream)
// and/or store T after an optional Snake (the NEXT conv's input activation, dac-vae/layers.py:22),
// so no elementwise kernel ever round-trips HBM between two convs.
#include <algorithm>
#include "common.h"
#include "gemm.h"
#include <utility>

template <typename T> struct Frag;
// LDS row pitch.  A fragment read is ds_read_b128 at (row l16, 16-byte chunk g), served in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): each group touches 16 different rows with two
// adjacent chunks.  Dense 64 B rows and 144 B rows both put half of those lanes on busy banks (8 LDS cycles per read
// instead of 4; rocprofv3 SQ_LDS_BANK_CONFLICT = 32 % of the GEMM's LDS cycles); pitches of 96 B and 160 B are conflict free.
template <> struct Frag<bf16_t> {
    static constexpr int CH = 8;              // elements per 16-byte chunk
    static constexpr int LDS_ROW = 48;        // 32 data + 16 pad elements: 96 B pitch
};
template <> struct Frag<float> {
    static constexpr int CH = 4;
    static constexpr int LDS_ROW = 40;        // 32 data + 8 pad elements: 160 B pitch
};

// NS > 1 is the split build (MMX_X2 / MMX_X3, include/mmx_hip.h): T (weights, LDS tiles, MFMA operands) is bf16, the A
// operand and out_act are fp32 in HBM (TA = float).  An A chunk of 4 floats is split into NS bf16 planes on its way
// into LDS (hi = bf16(x), then the rounded remainders), and every A x W fragment pair costs NS MFMAs into the same
// accumulator: exact bf16 x bf16 products, fp32 accumulation, i.e. 16 (NS = 2) or 24 (NS = 3) significant bits of the
// activation against exactly represented bf16 weights.
// NWP > 1 (MMX_X2W / MMX_X3W): the weights of an fp32 checkpoint, held as NWP bf16 PLANES hi + [mid +] lo = w (16 / 24 significant
// bits) side by side in every row: W[n][plane * (ldw / NWP) + k].  A product keeps every term A_s x W_p of order s + p < NS
// (NS = NWP = 2: hi*hi + hi*lo + lo*hi, 3 MFMAs; 3: 6 MFMAs): what is dropped is below the last kept bit of either operand.
template <typename T, typename TA, int NS, int BM, int BN, int WM, int WN, int NWP = 1>
__global__ __launch_bounds__(256) void gemm_win_kernel(GemmParams p) {
    constexpr int BK = 32;
    constexpr bool XS = NS > 1;
    static_assert(NWP == 1 || NWP == NS, "weight planes: as many as activation planes");
    static_assert(!XS || (sizeof(T) == 2 && sizeof(TA) == 4), "split build: bf16 weights, fp32 activations");
    static_assert(XS || sizeof(T) == sizeof(TA), "one storage type otherwise");
    constexpr int CH = Frag<T>::CH;
    constexpr int CPR = BK / CH;                       // W chunks per row per k-tile
    constexpr int CHA = 16 / sizeof(TA);               // A elements per 16-byte chunk
    constexpr int CPRA = BK / CHA;
    constexpr int LR = Frag<T>::LDS_ROW;
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;
    constexpr int A_CHUNKS = (BM * CPRA + 255) / 256, W_CHUNKS = (BN * CPR + 255) / 256;
    constexpr bool A_FULL = (BM * CPRA) % 256 == 0, W_FULL = (BN * CPR) % 256 == 0;
    constexpr bool PRECISE = sizeof(TA) == 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                // [NS][2][BM][LR]
    T* Ws = As + NS * 2 * BM * LR;                     // [NWP][2][BN][LR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const TA* A = reinterpret_cast<const TA*>(p.A) + (long)b * p.a_bstride;
    const T* W = reinterpret_cast<const T*>(p.W) + (long)b * p.w_bstride;

    // Per-thread chunk state.  Address generation is strength-reduced to a pointer bump per k-tile (the
    // straightforward form costs ~75 VALU instructions per 16-byte load and made the loop VALU-bound):
    // A: pointer + (tap, c) trackers, the pointer jumps by dil*lda - cin elements when c wraps into the next tap;
    // W: pointer += BK.  Row/column validity is decided from small integer trackers.
    // Loads go through buffer descriptors (base = this batch item's A / W, 2 GiB window): the per-lane part of an
    // address is a 32-bit byte offset, and a chunk that is padding simply gets an offset outside the window - the
    // hardware range check returns zeros.  No branch and no select around a load: a load under a branch makes the
    // compiler lose count of the loads in flight, and it then drains them all (s_waitcnt vmcnt(0)) before every LDS
    // store, which had collapsed the STAGES-deep prefetch to one k-tile (rocprofv3 PMC: waves parked 48 % of the time).
    constexpr unsigned OOB = 0x80000000u;
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TA*>(A), 0, 0x7fffffff, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, 0x7fffffff, 0x00020000);
    unsigned a_off[A_CHUNKS];
    int a_c[A_CHUNKS], a_tap[A_CHUNKS];
    long a_srow[A_CHUNKS];
    bool a_mok[A_CHUNKS];
    const unsigned tap_jump = (unsigned)(((long)p.dil * p.lda - p.cin) * (long)sizeof(TA));
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int id = tid + i * 256;
        const int r = id / CPRA, k = (id % CPRA) * CHA;
        const int m = m0 + r;
        a_tap[i] = k / p.cin;
        a_c[i] = k % p.cin;
        a_srow[i] = (long)m * p.row_stride + (long)a_tap[i] * p.dil + p.row_off;
        a_off[i] = (unsigned)((a_srow[i] * p.lda + a_c[i]) * (long)sizeof(TA));    // mod 2^32; used only when in range
        a_mok[i] = (m < p.M) && (A_FULL || id < BM * CPRA);
    }
    unsigned w_off[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) {
        const int id = tid + i * 256;
        const int r = id / CPR, kc = (id % CPR) * CH;
        const int n = n0 + r;
        const bool ok = (n < p.N) && (W_FULL || id < BN * CPR);
        w_off[i] = ok ? (unsigned)(((long)n * p.ldw + kc) * (long)sizeof(T)) : OOB;
    }
    // register-staged prefetch ring: STAGES k-tiles of global loads in flight per workgroup (these GEMMs are
    // short-K and latency bound: M ~ 500-1000 rows, K = 256..1024), 2 LDS buffers, one barrier per k-tile
    constexpr int STAGES = (sizeof(T) == 2 && !XS) ? 4 : 2;
    uint4 a_reg[STAGES][A_CHUNKS], w_reg[STAGES][NWP][W_CHUNKS];
    const unsigned w_plane = (unsigned)((p.ldw / NWP) * (long)sizeof(T));   // bytes between two planes of a weight row
    const int K = p.ntaps * p.cin;
    const int nk = (K + BK - 1) / BK;                  // W is zero padded to nk*BK columns (ldw >= nk*BK)

    auto load_tile = [&](int kt, auto slot_c) {       // must be called for kt = 0, 1, 2, ... in order
        constexpr int slot = decltype(slot_c)::value;
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const bool ok = a_mok[i] && (a_tap[i] < p.ntaps) && (a_srow[i] >= p.row_lo) && (a_srow[i] < p.row_hi);
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ok ? a_off[i] : OOB, 0, 0);
            a_reg[slot][i] = make_uint4(v[0], v[1], v[2], v[3]);
            a_off[i] += BK * (unsigned)sizeof(TA);
            a_c[i] += BK;
            while (a_c[i] >= p.cin) {
                a_c[i] -= p.cin;
                a_tap[i]++;
                a_srow[i] += p.dil;
                a_off[i] += tap_jump;
            }
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
#pragma unroll
            for (int p2 = 0; p2 < NWP; ++p2) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (kt < nk && w_off[i] != OOB) ? w_off[i] + p2 * w_plane : OOB, 0, 0);
                w_reg[slot][p2][i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            if (w_off[i] != OOB) w_off[i] += BK * (unsigned)sizeof(T);
        }
    };
    auto store_tile = [&](int stage, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            int id = tid + i * 256;
            if (A_FULL || id < BM * CPRA) {
                if constexpr (XS) {
                    const uint4 v = a_reg[slot][i];
                    float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) {
                        uint2 pk;
                        pk.x = pack_bf16x2(x[0], x[1]);
                        pk.y = pack_bf16x2(x[2], x[3]);
                        *reinterpret_cast<uint2*>(As + ((s2 * 2 + stage) * BM + id / CPRA) * LR + (id % CPRA) * CHA) = pk;
                        if (s2 + 1 < NS) {                // remainders (exact in fp32: the parts do not overlap)
                            x[0] -= __uint_as_float(pk.x << 16); x[1] -= __uint_as_float(pk.x & 0xffff0000u);
                            x[2] -= __uint_as_float(pk.y << 16); x[3] -= __uint_as_float(pk.y & 0xffff0000u);
                        }
                    }
                } else {
                    *reinterpret_cast<uint4*>(As + (stage * BM + id / CPRA) * LR + (id % CPRA) * CHA) = a_reg[slot][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
            int id = tid + i * 256;
            if (W_FULL || id < BN * CPR) {
#pragma unroll
                for (int p2 = 0; p2 < NWP; ++p2)
                    *reinterpret_cast<uint4*>(Ws + ((p2 * 2 + stage) * BN + id / CPR) * LR + (id % CPR) * CH) = w_reg[slot][p2][i];
            }
        }
    };

    float4_t acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int st) {
        const T* as = As + (st * BM + wm * (BM / WM) + l16) * LR;
        const T* ws = Ws + (st * BN + wn * (BN / WN) + l16) * LR;
        if constexpr (sizeof(T) == 2) {
            short8_t bfr[NWP][NF];
#pragma unroll
            for (int p2 = 0; p2 < NWP; ++p2)
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[p2][j] = *reinterpret_cast<const short8_t*>(ws + p2 * 2 * BN * LR + j * 16 * LR + 8 * g);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {          // the planes of the split A tile (one plane otherwise)
                short8_t af[MF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *reinterpret_cast<const short8_t*>(as + s2 * 2 * BM * LR + i * 16 * LR + 8 * g);
#pragma unroll
                for (int p2 = 0; p2 < NWP; ++p2) {
                    if (s2 + p2 >= NS) continue;       // (compile time) terms below the last kept bit
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[p2][j], acc[i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {              // two 16-deep chunks; step s multiplies k = 16c + 4g' + s
                float4_t af[MF], bfr[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i) af[i] = *reinterpret_cast<const float4_t*>(as + i * 16 * LR + 16 * c + 4 * g);
#pragma unroll
                for (int j = 0; j < NF; ++j) bfr[j] = *reinterpret_cast<const float4_t*>(ws + j * 16 * LR + 16 * c + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bfr[j][s], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto for_each_slot = [&](auto&& fn) {
        [&]<int... S>(std::integer_sequence<int, S...>) { (fn(std::integral_constant<int, S>{}), ...); }
        (std::make_integer_sequence<int, STAGES>{});
    };
    // prologue: tiles 0 .. STAGES-1 in flight (load_tile advances the (tap, c) trackers in tile order; a tile index
    // >= nk loads nothing: its offsets are out of the window).  The steady-state loop is branch free, every iteration
    // issues and retires the same number of loads, so the compiler can wait for exactly the tile it stores
    // (s_waitcnt vmcnt(2*(STAGES-1)) for 64x64) instead of draining the ring.
    for_each_slot([&](auto sc) { load_tile(decltype(sc)::value, sc); });
    int kt0 = 0;
    for (; kt0 + STAGES <= nk; kt0 += STAGES) {
        for_each_slot([&](auto sc) {
            const int kt = kt0 + decltype(sc)::value;
            // LDS[kt&1] was last read by compute(kt-2); every wave passed barrier(kt-1) after finishing it
            store_tile(decltype(sc)::value & 1, sc);
            load_tile(kt + STAGES, sc);
            __syncthreads();
            compute(decltype(sc)::value & 1);
        });
    }
    for_each_slot([&](auto sc) {                       // the last nk % STAGES tiles (already in flight)
        const int kt = kt0 + decltype(sc)::value;
        if (kt < nk) {                                 // uniform
            store_tile(decltype(sc)::value & 1, sc);
            __syncthreads();
            compute(decltype(sc)::value & 1);
        }
    });

    // ------------------------------------------------------------------ fused epilogue
    // The MFMA C layout gives a lane 4 ROWS x 1 column per fragment: storing from it writes 32-64 B row
    // segments (measured: ~250 GB/s of output, the whole cost of a short-K GEMM).  Instead each wave passes its
    // tile through a private LDS patch, 16 rows at a time, and every lane then owns CW consecutive columns of one
    // row: residual loads and both output stores are 16-byte accesses, 4 lanes cover a full 128-256 B line.
    __syncthreads();                                   // all waves are done with the operand tiles
    constexpr int WNC = BN / WN;                       // columns of the wave tile
    constexpr int CW = WNC / 4;                        // consecutive columns per lane (16 / 8 / 4)
    constexpr int LDC = WNC + 4;                       // padded fp32 row: conflict-free ds_write_b32
    float* Cs = reinterpret_cast<float*>(smem) + wave * 16 * LDC;
    float* outf = p.out_f32 ? p.out_f32 + (long)b * p.of_bstride : nullptr;
    TA* outa = p.out_act ? reinterpret_cast<TA*>(p.out_act) + (long)b * p.oa_bstride : nullptr;
    const float* res = p.residual ? p.residual + (long)b * p.r_bstride : nullptr;
    const float* rmask = p.rowmask ? p.rowmask + (long)b * p.rm_bstride : nullptr;
    constexpr int VA = 16 / sizeof(TA);                // act elements per 16 bytes
    const bool vec_f = outf && (p.ldo_f % 4 == 0) && (p.out_off % 4 == 0) && (p.of_bstride % 4 == 0) && ((uintptr_t)p.out_f32 % 16 == 0);
    const bool vec_a = outa && (p.ldo_a % VA == 0) && (p.out_off % VA == 0) && (p.oa_bstride % VA == 0) && ((uintptr_t)p.out_act % 16 == 0) && (CW % VA == 0);
    const bool vec_r = res && (p.ldr % 4 == 0) && (p.r_bstride % 4 == 0) && ((uintptr_t)p.residual % 16 == 0);
    const int erow = lane >> 2, ecol = (lane & 3) * CW;
    auto epilogue = [&](auto act_k, auto act2_k) {
    constexpr int ACT = decltype(act_k)::value, ACT2 = decltype(act2_k)::value;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(4 * g + r) * LDC + j * 16 + l16] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float v[CW];
#pragma unroll
        for (int c = 0; c < CW; c += 4) {
            float4_t t4 = *reinterpret_cast<const float4_t*>(Cs + erow * LDC + ecol + c);
            v[c] = t4[0]; v[c + 1] = t4[1]; v[c + 2] = t4[2]; v[c + 3] = t4[3];
        }
        __builtin_amdgcn_wave_barrier();               // patch is rewritten by the next i
        const int m = m0 + wm * (BM / WM) + i * 16 + erow;
        const int nb = n0 + wn * WNC + ecol;
        if (m >= p.M || nb >= p.N) continue;
        const bool full = nb + CW <= p.N;
        const float rm = rmask ? rmask[m] : 1.f;
        float rr[CW];
        if (res) {
            if (vec_r && full) {
#pragma unroll
                for (int c = 0; c < CW; c += 4) {
                    float4_t t4 = *reinterpret_cast<const float4_t*>(res + (long)m * p.ldr + nb + c);
                    rr[c] = t4[0]; rr[c + 1] = t4[1]; rr[c + 2] = t4[2]; rr[c + 3] = t4[3];
                }
            } else {
#pragma unroll
                for (int c = 0; c < CW; ++c) rr[c] = (nb + c < p.N) ? res[(long)m * p.ldr + nb + c] : 0.f;
            }
        }
        int bidx = p.bias ? (p.bias_per_row ? m : nb % p.bias_mod) : 0;
        int aidx = p.alpha ? nb % p.alpha_mod : 0;
        float w2[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            float x = v[c];
            if (p.bias) {
                x += p.bias[bidx];
                if (!p.bias_per_row && ++bidx == p.bias_mod) bidx = 0;
            }
            x = act_c<ACT, PRECISE>(x, p.slope);
            if (res) x += rr[c];
            x *= rm;
            v[c] = x;
            if (outa) {
                float w = act_c<ACT2, PRECISE>(x, p.slope);
                if (p.alpha) {
                    w = snake_apply<PRECISE>(w, p.alpha[aidx]);
                    if (++aidx == p.alpha_mod) aidx = 0;
                }
                w2[c] = w;
            }
        }
        if (outf) {
            const long lin = (long)m * p.ldo_f + nb + p.out_off;
            if (vec_f && full && lin >= 0 && lin + CW <= p.out_len) {
#pragma unroll
                for (int c = 0; c < CW; c += 4)
                    *reinterpret_cast<float4_t*>(outf + lin + c) = float4_t{v[c], v[c + 1], v[c + 2], v[c + 3]};
            } else {
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (nb + c < p.N && lin + c >= 0 && lin + c < p.out_len) outf[lin + c] = v[c];
            }
        }
        if (outa) {
            const long lin = (long)m * p.ldo_a + nb + p.out_off;
            if (vec_a && full && lin >= 0 && lin + CW <= p.out_len) {
                if constexpr (sizeof(TA) == 2) {
#pragma unroll
                    for (int c = 0; c < CW; c += 8) {
                        uint4 pk;
                        pk.x = pack_bf16x2(w2[c], w2[c + 1]);
                        pk.y = pack_bf16x2(w2[c + 2], w2[c + 3]);
                        pk.z = pack_bf16x2(w2[c + 4], w2[c + 5]);
                        pk.w = pack_bf16x2(w2[c + 6], w2[c + 7]);
                        *reinterpret_cast<uint4*>(outa + lin + c) = pk;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CW; c += 4)
                        *reinterpret_cast<float4_t*>(outa + lin + c) = float4_t{w2[c], w2[c + 1], w2[c + 2], w2[c + 3]};
                }
            } else {
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (nb + c < p.N && lin + c >= 0 && lin + c < p.out_len) outa[lin + c] = Cvt<TA>::from_f(w2[c]);
            }
        }
    }
    };
    using std::integral_constant;
    switch (p.act) {                                   // uniform: one activation per launch
        case ACT_LRELU: epilogue(integral_constant<int, ACT_LRELU>{}, integral_constant<int, ACT_NONE>{}); break;
        case ACT_GELU: epilogue(integral_constant<int, ACT_GELU>{}, integral_constant<int, ACT_NONE>{}); break;
        case ACT_SILU: epilogue(integral_constant<int, ACT_SILU>{}, integral_constant<int, ACT_NONE>{}); break;
        case ACT_MISH: epilogue(integral_constant<int, ACT_MISH>{}, integral_constant<int, ACT_NONE>{}); break;
        case ACT_TANH: epilogue(integral_constant<int, ACT_TANH>{}, integral_constant<int, ACT_NONE>{}); break;
        default:
            if (p.act2 == ACT_MISH) epilogue(integral_constant<int, ACT_NONE>{}, integral_constant<int, ACT_MISH>{});
            else epilogue(integral_constant<int, ACT_NONE>{}, integral_constant<int, ACT_NONE>{});
    }
}

template <typename T, typename TA, int NS, int BM, int BN, int WM, int WN, int NWP = 1>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.batch);
    size_t lds = (size_t)2 * (NS * BM + NWP * BN) * Frag<T>::LDS_ROW * sizeof(T);
    const size_t lds_epi = (size_t)4 * 16 * (BN / WN + 4) * sizeof(float);    // per-wave epilogue patches
    if (lds_epi > lds) lds = lds_epi;
    MMX_LDS_OPT_IN((gemm_win_kernel<T, TA, NS, BM, BN, WM, WN, NWP>), lds);
    hipLaunchKernelGGL((gemm_win_kernel<T, TA, NS, BM, BN, WM, WN, NWP>), grid, dim3(256), lds, s, p);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}

template <typename T, typename TA, int NS, int NWP = 1>
static int launch_T(const GemmParams& p, hipStream_t s, int force_tile) {
    // shape validation the kernel relies on (16-byte chunk loads)
    constexpr int CH = Frag<T>::CH;
    constexpr int CHA = 16 / sizeof(TA);
    MMX_CHECK_ARG(p.A && p.W && p.M > 0 && p.N > 0 && p.batch > 0 && p.ntaps >= 1);
    MMX_CHECK_ARG(p.cin % CH == 0 && p.lda % CHA == 0 && p.ldw % CH == 0);
    MMX_CHECK_ARG(p.a_bstride % CHA == 0 && p.w_bstride % CH == 0);
    MMX_CHECK_ARG(p.ldw % NWP == 0 && (p.ldw / NWP) % CH == 0 && p.ldw / NWP >= ((p.ntaps * p.cin + 31) / 32) * 32);
    MMX_CHECK_ARG(p.bias_mod > 0 && p.alpha_mod > 0 && p.row_stride >= 1);
    MMX_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0);
    MMX_CHECK_ARG(p.out_f32 || p.out_act);
    // operands are addressed with 32-bit byte offsets inside a 2 GiB buffer window per batch item
    const long a_rows = std::min<long>(p.row_hi, (long)(p.M - 1) * p.row_stride + (long)(p.ntaps - 1) * p.dil + p.row_off + 1);
    MMX_CHECK_ARG((double)a_rows * (double)p.lda * sizeof(TA) < 2147483000.0 &&
                  (double)p.N * (double)p.ldw * sizeof(T) < 2147483000.0);
    MMX_CHECK_ARG(p.act2 == ACT_NONE || (p.act2 == ACT_MISH && p.act == ACT_NONE));   // the only fused pair in use
    // largest tile that still gives every CU a workgroup (256 CUs); short-K GEMMs want many MFMAs per barrier
    auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.batch; };
    if constexpr (NWP > 1) {                           // weight planes: two tiles (LDS: 4 or 6 operand planes per k-tile)
        if (force_tile != 0 && force_tile != MMX_TILE_64x64 && force_tile != MMX_TILE_32x64) return MMX_EARG;
        if (force_tile != MMX_TILE_32x64 && (force_tile == MMX_TILE_64x64 || blocks(64, 64) >= 64 || p.M > 32))
            return launch_cfg<T, TA, NS, 64, 64, 2, 2, NWP>(p, s);
        return launch_cfg<T, TA, NS, 32, 64, 1, 4, NWP>(p, s);
    }
    switch (force_tile) {                              // mmx_gemm_win_tile: the tuning entry (tools/microbench.py)
        case 0: break;
        case MMX_TILE_128x128: return launch_cfg<T, TA, NS, 128, 128, 2, 2>(p, s);
        case MMX_TILE_128x64: return launch_cfg<T, TA, NS, 128, 64, 4, 1>(p, s);
        case MMX_TILE_64x64: return launch_cfg<T, TA, NS, 64, 64, 2, 2>(p, s);
        case MMX_TILE_32x64: return launch_cfg<T, TA, NS, 32, 64, 1, 4>(p, s);
        default: return MMX_EARG;
    }
    // measured on MI355X (tools/microbench.py tiles, profiles/r01_gemm_tiles.txt): these GEMMs are short-K and
    // latency bound, so more (smaller) workgroups win until the problem is large: 64x64 beats 128x128 up to
    // ~1000 128-tiles (234 vs 239 TFLOP/s at M=8192,N=1024,K=256; 256 vs 153 at N=256,K=1024) and by 2x at M=1024.
    const long b128 = blocks(128, 128);
    if (p.N > 64 && (b128 >= 1024 || (p.ntaps * p.cin >= 1024 && b128 >= 512))) return launch_cfg<T, TA, NS, 128, 128, 2, 2>(p, s);
    if (p.N <= 64 && blocks(128, 64) >= 512) return launch_cfg<T, TA, NS, 128, 64, 4, 1>(p, s);
    if (blocks(64, 64) >= 64 || p.M > 32) return launch_cfg<T, TA, NS, 64, 64, 2, 2>(p, s);
    return launch_cfg<T, TA, NS, 32, 64, 1, 4>(p, s);
}

extern "C" int mmx_gemm_win_tile(const GemmParams* p, int dtype, int tile, hipStream_t stream) {
    MMX_CHECK_ARG(p != nullptr);
    if (dtype == MMX_BF16) return launch_T<bf16_t, bf16_t, 1>(*p, stream, tile);
    if (dtype == MMX_F32) return launch_T<float, float, 1>(*p, stream, tile);
    if (dtype == MMX_X2) return launch_T<bf16_t, float, 2>(*p, stream, tile);
    if (dtype == MMX_X3) return launch_T<bf16_t, float, 3>(*p, stream, tile);
    if (dtype == MMX_X2W) return launch_T<bf16_t, float, 2, 2>(*p, stream, tile);
    if (dtype == MMX_X3W) return launch_T<bf16_t, float, 3, 3>(*p, stream, tile);
    return MMX_EARG;
}
extern "C" int mmx_gemm_win(const GemmParams* p, int dtype, hipStream_t stream) {
    return mmx_gemm_win_tile(p, dtype, 0, stream);
}
