// Device-resident token sampler + AR-loop bookkeeping: one workgroup per sequence, no host round trip
// per token (the reference pays one .item() sync per token, llm.py:752).  Restates
//   llm.py:751        logp = log_softmax(logits)
//   common.py:111-139 ras_sampling -> nucleus_sampling (softmax, stable sort desc, cum < top_p, n < top_k,
//                     multinomial) and random_sampling on repetition within the last win_size tokens
//   llm.py:259-274    sampling_ids: re-draw (<= 100 trials) while EOS is drawn and ignore_eos
//   llm.py:753-760    EOS stops; ids > EOS are skipped WITHOUT updating lm_input; otherwise append + embed
// torch.multinomial(p,1) == argmax(p / e), e ~ Exp(1): the draws come from Philox4x32-10 exactly as
// oracle/philox.py defines them, so the CPU oracle and this kernel consume identical noise.
#include "common.h"
#include "../../include/mmx_hip.h"

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float exp_noise(unsigned long long seed, unsigned seq, unsigned step, unsigned trial, unsigned which,
                                           unsigned i) {
    unsigned w[4];
    philox4x32_10(i >> 2, 2u * trial + which, step, seq, (unsigned)seed, (unsigned)(seed >> 32), w);
    const float u = ((float)(w[i & 3] >> 8) + 0.5f) * 5.9604644775390625e-08f;   // 2^-24
    return -logf(u);
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {     // larger value, then smaller index (stable sort order)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax x, ArgMax* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax y;
        y.v = __shfl_xor(x.v, o, 64);
        y.i = __shfl_xor(x.i, o, 64);
        x = better(x, y);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    ArgMax r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = better(r, sh[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    return r;
}

// 512 threads = 8 waves per sequence (two per SIMD): 13 logits per thread.  The kernel is ONE dependent chain per decode step
// (it sits between the head projection and the next step's first launch), so what counts is its length, not its work:
//   * the threshold of the candidate list comes from 64 GROUP maxima (8 consecutive lanes each, reduced with DPP), ranked by
//     one wave - 64 x 64 comparisons instead of 256 x 256;
//   * list ranks are counted by 8 lanes per entry (DPP sum), the running sum of the sorted prefix walks registers of one
//     wave (v_readlane) instead of 25 dependent LDS reads, the nucleus draw is reduced inside wave 0;
//   * the repetition window is read by 10 lanes at once.
// Round 3's kernel (256 threads, 26 logits per thread, every thread ranking 256 local maxima) took 29 us per step.
constexpr int SAMP_THREADS = 512;
constexpr int SAMP_WAVES = SAMP_THREADS / 64;
constexpr int SAMP_MAXV = 13;        // V <= 6656 (speech_token_size + 3 = 6564)
constexpr int SAMP_MAXK = 64;
constexpr int SAMP_GROUPS = SAMP_THREADS / 8;   // 64 group maxima
constexpr int SAMP_LIST = 512;       // candidates >= the top_k-th group maximum (ties / clustered values included)

// (value, index) of the better of two candidates across lanes with a DPP control (see common.h dpp_f32)
template <int CTRL>
__device__ __forceinline__ ArgMax dpp_better(ArgMax a) {
    ArgMax o;
    o.v = dpp_f32<CTRL>(a.v);
    o.i = __builtin_amdgcn_update_dpp(0, a.i, CTRL, 0xf, 0xf, false);
    return better(a, o);
}

__global__ __launch_bounds__(SAMP_THREADS) void sample_step_kernel(
    const float* __restrict__ logits, long ldl, int V, int eos_id, int top_k, float top_p, int win_size, float tau_r,
    unsigned long long seed, int32_t* __restrict__ state, int32_t* __restrict__ out_tokens, int max_out,
    int32_t* __restrict__ sampled, const int32_t* __restrict__ forced, const float* __restrict__ speech_emb, int E,
    float* __restrict__ next_x, long ldx, float* __restrict__ logp_out) {
    __shared__ float shf[SAMP_WAVES];
    __shared__ ArgMax sha[SAMP_WAVES];
    __shared__ float cand_p[SAMP_MAXK];
    __shared__ int cand_i[SAMP_MAXK];
    __shared__ int sh_top, sh_cnt, sh_thr_i;
    __shared__ float sh_thr_v;
    __shared__ __attribute__((aligned(16))) float gmax_p[SAMP_GROUPS];
    __shared__ __attribute__((aligned(16))) int gmax_i[SAMP_GROUPS];
    __shared__ float p_lds[SAMP_THREADS * SAMP_MAXV];
    __shared__ float list_p[SAMP_LIST];
    __shared__ int list_i[SAMP_LIST];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = gridDim.x;                          // state is field-major: state[field * B + b]
    int32_t* st = state + b;
#define ST(f) st[(f) * nb]
    const int pos = ST(0), step = ST(1), n_out = ST(2), finished = ST(3), min_len = ST(4), max_len = ST(5), seq = ST(6);
    if (finished) return;                              // uniform per block
    const float* lg = logits + (long)b * ldl;
    // the repetition window (common.py:113: the last win_size accepted tokens), one token per lane, requested with the logits
    const int wn = min(win_size, n_out);
    int hist = -1;
    if (lane < wn) hist = out_tokens[(long)b * max_out + n_out - wn + lane];

    // log_softmax, then softmax of it (the reference's two stages, common.py:122)
    float x[SAMP_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SAMP_MAXV; ++i) {
        int idx = tid + i * SAMP_THREADS;
        x[i] = idx < V ? lg[idx] : -INFINITY;
        mx = fmaxf(mx, x[i]);
    }
    mx = block_max(mx, shf);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_MAXV; ++i) se += (tid + i * SAMP_THREADS < V) ? expf(x[i] - mx) : 0.f;
    se = block_sum(se, shf);
    const float lse = logf(se);
#pragma unroll
    for (int i = 0; i < SAMP_MAXV; ++i) {
        int idx = tid + i * SAMP_THREADS;
        if (idx < V) {
            x[i] = (x[i] - mx) - lse;                  // logp
            if (logp_out) logp_out[(long)b * V + idx] = x[i];
        }
    }
    // max(logp) without another block reduction: the maximal logit gives (mx - mx) - lse = 0.0f - lse exactly, and every
    // other element's (x - mx) is <= 0, so the maximum of the values just computed IS this one
    const float mx2 = 0.0f - lse;
    float se2 = 0.f;
#pragma unroll
    for (int i = 0; i < SAMP_MAXV; ++i) {
        int idx = tid + i * SAMP_THREADS;
        x[i] = idx < V ? expf(x[i] - mx2) : -1.f;      // unnormalised p (padding -1 never wins)
        se2 += idx < V ? x[i] : 0.f;
    }
    se2 = block_sum(se2, shf);
#pragma unroll
    for (int i = 0; i < SAMP_MAXV; ++i)
        if (tid + i * SAMP_THREADS < V) {
            x[i] = x[i] / se2;                         // p
            p_lds[tid + i * SAMP_THREADS] = x[i];
        }

    // nucleus candidates = prefix of the stable descending sort (value desc, index asc), found by RANK COUNTING
    // rather than by repeated arg-max rounds: (1) the maximum of every group of 8 lanes (104 elements); (2) one wave counts,
    // for each of the 64 group maxima, how many of the others precede it in the sort order - the one with rank top_k-1 is a
    // threshold T with at least top_k elements >= T, so every global top_k element is >= T; (3) all elements >= T are pushed
    // to a small LDS list; (4) each list entry's rank inside the list is its position in the sorted prefix; (5) the fp32
    // running sum walks that prefix in the reference's order (common.py:124-131).  Ranks are unique because indices are.
    auto precedes = [](float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); };
    {
        ArgMax a{-2.f, 0x7fffffff};
#pragma unroll
        for (int i = 0; i < SAMP_MAXV; ++i) a = better(a, ArgMax{x[i], tid + i * SAMP_THREADS});
        a = dpp_better<0xB1>(a);                       // lanes xor 1, xor 2, then the other quad of the 8
        a = dpp_better<0x4E>(a);
        a = dpp_better<0x141>(a);
        if ((lane & 7) == 0) { gmax_p[tid >> 3] = a.v; gmax_i[tid >> 3] = a.i; }
    }
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    if (wave == 0) {
        const float mv = gmax_p[lane];
        const int mi = gmax_i[lane];
        int rk = 0;
#pragma unroll
        for (int j = 0; j < SAMP_GROUPS; j += 4) {     // 16-byte broadcast reads
            const float4 pv = *reinterpret_cast<const float4*>(gmax_p + j);
            const int4 iv = *reinterpret_cast<const int4*>(gmax_i + j);
            rk += (precedes(pv.x, iv.x, mv, mi) ? 1 : 0) + (precedes(pv.y, iv.y, mv, mi) ? 1 : 0) +
                  (precedes(pv.z, iv.z, mv, mi) ? 1 : 0) + (precedes(pv.w, iv.w, mv, mi) ? 1 : 0);
        }
        if (rk == min(top_k, SAMP_GROUPS) - 1) { sh_thr_v = mv; sh_thr_i = mi; }
    }
    __syncthreads();
    {
        const ArgMax thr{sh_thr_v, sh_thr_i};
#pragma unroll
        for (int i = 0; i < SAMP_MAXV; ++i) {
            const ArgMax e{x[i], tid + i * SAMP_THREADS};
            // e >= thr in the sort order  <=>  !(thr strictly precedes e)
            const bool take = e.i < V && !precedes(thr.v, thr.i, e.v, e.i);
            // one LDS atomic per wave and round instead of one per taker (the list's order is irrelevant: ranks are
            // recomputed from (value, index) below)
            const unsigned long long tm = __ballot(take);
            if (tm) {
                const int leader = __ffsll((long long)tm) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&sh_cnt, __popcll(tm));
                base = __shfl(base, leader, 64);
                if (take) {
                    const int slot = base + __popcll(tm & ((1ull << lane) - 1ull));
                    if (slot < SAMP_LIST) { list_p[slot] = e.v; list_i[slot] = e.i; }
                }
            }
        }
    }
    __syncthreads();
    const int cnt = min(sh_cnt, SAMP_LIST);
    // rank of list entry `id` inside the list: 8 lanes share the comparisons of one entry (64 entries per round)
    for (int id0 = 0; id0 < cnt; id0 += SAMP_THREADS / 8) {
        const int id = id0 + (tid >> 3), part = tid & 7;
        const bool live = id < cnt;
        const float mv = live ? list_p[id] : 0.f;
        const int mi = live ? list_i[id] : 0;
        int rk = 0;
        for (int j = part; j < cnt; j += 8) rk += precedes(list_p[j], list_i[j], mv, mi) ? 1 : 0;
        rk += __builtin_amdgcn_update_dpp(0, rk, 0xB1, 0xf, 0xf, false);
        rk += __builtin_amdgcn_update_dpp(0, rk, 0x4E, 0xf, 0xf, false);
        rk += __builtin_amdgcn_update_dpp(0, rk, 0x141, 0xf, 0xf, false);
        if (live && part == 0 && rk < SAMP_MAXK) { cand_p[rk] = mv; cand_i[rk] = mi; }
    }
    __syncthreads();
    // every wave walks the sorted prefix on its own (no barrier, no broadcast): lane i holds candidate i, the fp32 running sum
    // adds them in the reference's order (common.py:127)
    const int lim = min(min(top_k, SAMP_MAXK), cnt);
    const float cp = lane < lim ? cand_p[lane] : 0.f;
    const int ci = lane < lim ? cand_i[lane] : 0;
    int nc = 0;
    {
        float cum = 0.f;
#pragma unroll 1
        while (cum < top_p && nc < lim) {
            cum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cp), nc));
            nc++;
        }
    }

    const bool ignore_eos = step < min_len;
    int top = 0;
    for (int trial = 0;; ++trial) {
        // nucleus draw: argmax_i cand_p[i] / e_i, reduced inside every wave (the same values in each: no barrier)
        ArgMax a{-1.f, 0x7fffffff};
        if (lane < nc) a = ArgMax{cp / exp_noise(seed, seq, step, trial, 0, lane), lane};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ArgMax y;
            y.v = __shfl_xor(a.v, o, 64);
            y.i = __shfl_xor(a.i, o, 64);
            a = better(a, y);
        }
        top = __builtin_amdgcn_readlane(ci, a.i);
        // repetition-aware fallback (common.py:113-115)
        const int rep = __popcll(__ballot(hist == top));
        if ((float)rep >= (float)win_size * tau_r) {
            ArgMax r{-1.f, 0x7fffffff};
#pragma unroll 1
            for (int idx = tid; idx < V; idx += SAMP_THREADS)      // rolled: p from LDS keeps this rare path compact
                r = better(r, ArgMax{p_lds[idx] / exp_noise(seed, seq, step, trial, 1, idx), idx});
            r = block_argmax(r, sha);
            top = r.i;
        }
        if (!ignore_eos || top != eos_id) break;
        if (trial >= 100) { if (tid == 0) ST(7) = 1; break; }    // llm.py:271-273 raises here; flag + accept
    }
    if (tid == 0 && sampled) sampled[(long)b * max_out + step] = top;
    if (forced) top = forced[(long)b * max_out + step];         // uniform: every thread reads the same word
    if (top == eos_id) {
        if (tid == 0) { ST(3) = 1; ST(1) = step + 1; }
        return;
    }
    if (top < eos_id) {
        for (int c = tid; c < E; c += SAMP_THREADS) next_x[(long)b * ldx + c] = speech_emb[(long)top * E + c];
        if (tid == 0) { out_tokens[(long)b * max_out + n_out] = top; ST(2) = n_out + 1; }
    }
    if (tid == 0) {
        ST(0) = pos + 1;
        ST(1) = step + 1;
        if (step + 1 >= max_len) ST(3) = 1;              // llm.py:746: for i in range(max_len)
    }
#undef ST
}

extern "C" int mmx_sample_step(const float* logits, int64_t ldl, int V, int B, int eos_id, int top_k, float top_p,
                               int win_size, float tau_r, uint64_t seed, int32_t* state, int32_t* out_tokens, int max_out,
                               int32_t* sampled, const int32_t* forced, const float* speech_emb, int E, float* next_x,
                               int64_t ldx, float* logp_out, hipStream_t stream) {
    MMX_CHECK_ARG(logits && state && out_tokens && speech_emb && next_x && B > 0 && V > 0 && V <= SAMP_THREADS * SAMP_MAXV);
    MMX_CHECK_ARG(top_k > 0 && top_k <= SAMP_MAXK && top_k <= SAMP_GROUPS && win_size >= 0 && win_size <= 64 && max_out > 0 && E > 0 && eos_id < V);
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(SAMP_THREADS), 0, stream, logits, ldl, V, eos_id, top_k, top_p, win_size,
                       tau_r, (unsigned long long)seed, state, out_tokens, max_out, sampled, forced, speech_emb, E, next_x, ldx, logp_out);
    MMX_LAUNCH_CHECK();
    return MMX_OK;
}
