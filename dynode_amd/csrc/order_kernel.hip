// order_kernel.hip -- dispatch order of a batch from a cost model (dyn_cost_order).
//
// The solve kernels integrate the 2..32 trajectories of a wave in lock-step (an iteration lasts until the slowest group
// has stepped, a save round until the group with the most rows is done) and the hardware hands waves to SIMDs in index
// order.  Which trajectories share a wave and which waves start last therefore changes the launch time by 10-15 % (measured:
// DESIGN.md section 9) although it cannot change a single output bit.  diffrax under vmap has the same property on a CPU and
// nobody orders the batch there; here it is two tiny launches in front of the solve:
//   cost_keys        every trajectory: predicted number of step attempts = a quadratic form in the standardised logarithms
//                    of its (varying, positive) parameters -- coefficients fitted by the host from step counts the solve
//                    kernels returned on earlier batches (dynode_amd/schedule.py) -- quantised to a bucket, most expensive first
//   order_from_keys  one workgroup: counting sort of the buckets (histogram in LDS, scan, scatter)
// The order of equal keys is whatever the LDS atomics produce; the outputs of the solve do not depend on it.
#include "../../include/dynode_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

namespace dynord {

constexpr int kBuckets = 4096;

// Exchangeable strains (n_sym = S > 1): the first sym_blocks * S parameters are blocks [quantity][strain] of a model that
// treats its strains alike (beta, gamma, sigma, omega of dyn_model_desc family 0), so the step count is a symmetric function
// of the strains and the forecast is much sharper on a canonical labelling: strains sorted by block 0 / block 1 (= r0),
// largest first.  Parameter c < sym_blocks * S is then read at [c / S][rank (c % S)].
constexpr int kMaxSym = 8;

// NF = compiled feature capacity (the host pads the coefficient arrays of a smaller model with zeros): with the loops
// unrolled the features stay in registers and the coefficients -- wave-uniform addresses -- arrive through scalar loads, so a
// trajectory costs its NF logarithms and NF (NF + 3) / 2 FMAs.
template <typename T, int NF>
__global__ void __launch_bounds__(256) cost_keys(const T *__restrict__ params, int64_t B, int P,
                                                 const int32_t *__restrict__ cols, const float *__restrict__ coef,
                                                 float key_scale, int n_sym, int sym_blocks, int32_t *__restrict__ keys) {
    constexpr int NQ = 1 + NF + NF * (NF + 1) / 2;      // quadratic form, then centre and 1 / spread of every feature
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const T *p = params + b * P;
    int rank[kMaxSym];
    if (n_sym > 1) { // insertion sort of the strains by block 0 / block 1, descending, stable
        float key[kMaxSym];
#pragma unroll
        for (int sI = 0; sI < kMaxSym; ++sI) {
            if (sI < n_sym) {
                const float k = (float)p[sI] / (float)p[n_sym + sI];
                int at = sI;
#pragma unroll
                for (int back = kMaxSym - 1; back > 0; --back) {      // (unrolled: `key` and `rank` stay in registers)
                    if (back <= sI && at == back && key[back - 1] < k) {
                        key[back] = key[back - 1];
                        rank[back] = rank[back - 1];
                        at = back - 1;
                    }
                }
#pragma unroll
                for (int q = 0; q < kMaxSym; ++q)
                    if (q == at) {
                        key[q] = k;
                        rank[q] = sI;
                    }
            }
        }
    }
    float l[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) { // column c >= 0: the logarithm of parameter c; c < 0: parameter -(c + 1) itself
        const int cc = cols[i];
        int src = cc < 0 ? -(cc + 1) : cc;
        if (n_sym > 1 && src < sym_blocks * n_sym) {
            const int sI = src % n_sym;
            int r = 0;
#pragma unroll
            for (int q = 0; q < kMaxSym; ++q) r = q == sI ? rank[q] : r;
            src = (src / n_sym) * n_sym + r;
        }
        const float v = (float)p[src];
        l[i] = ((cc < 0 ? v : __logf(fmaxf(v, 1e-30f))) - coef[NQ + i]) * coef[NQ + NF + i];
    }
    float pred = coef[0];
    int q = 1 + NF;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        float row = coef[1 + i];                  // a_i + sum_{j >= i} q_ij l_j
#pragma unroll
        for (int j = i; j < NF; ++j) row = fmaf(coef[q++], l[j], row);
        pred = fmaf(row, l[i], pred);
    }
    // non-finite predictions (NaN parameters) go last: they fail at once
    int bucket = (pred == pred) ? (int)fminf(fmaxf(pred * key_scale, 0.0f), (float)(kBuckets - 1)) : 0;
    keys[b] = kBuckets - 1 - bucket;           // ascending key = descending cost
}

// One workgroup: counting sort of the bucket keys (histogram in LDS, scan, scatter).  A thread keeps up to PER_THREAD of its
// keys in registers between the two passes; longer batches read them again.
// `pair` > 0 (a launch whose waves all start at once: one residency round): the sorted list is cut into waves of `pair`
// trajectories and the waves are dealt heavy, light, heavy, light ... (wave w of the descending order goes to slot 2 w in the
// first half and 2 (n - 1 - w) + 1 in the second), so that the waves sharing a SIMD add up to about the same work.  With
// several rounds the descending order itself (longest first) is the better one: measured, DESIGN.md section 9.
__device__ inline int64_t dealt(int64_t pos, int64_t B, int pair) {
    if (pair <= 0) return pos;
    const int64_t n = B / pair, w = pos / pair, r = pos % pair;      // (B is a multiple of `pair`: checked by the host side)
    const int64_t slot = w < (n + 1) / 2 ? 2 * w : 2 * (n - 1 - w) + 1;
    return slot * pair + r;
}

__global__ void __launch_bounds__(1024) order_from_keys(const int32_t *__restrict__ keys, int64_t B, int32_t *__restrict__ order,
                                                        int pair) {
    __shared__ int32_t hist[kBuckets];
    __shared__ int32_t part[1024];
    constexpr int PER_THREAD = 16;
    const int t = threadIdx.x;
    for (int i = t; i < kBuckets; i += 1024) hist[i] = 0;
    int32_t mine[PER_THREAD];
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
        const int64_t b = t + (int64_t)i * 1024;
        mine[i] = b < B ? keys[b] : -1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i)
        if (mine[i] >= 0) atomicAdd(&hist[mine[i]], 1);
    for (int64_t b = t + (int64_t)PER_THREAD * 1024; b < B; b += 1024) atomicAdd(&hist[keys[b]], 1);
    __syncthreads();
    // exclusive scan of 4096 buckets: 4 per thread, then the 1024 partial sums
    constexpr int PER = kBuckets / 1024;
    int32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = hist[t * PER + i];
        sum += v[i];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int32_t add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int32_t base = part[t] - sum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        hist[t * PER + i] = base;
        base += v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i)
        if (mine[i] >= 0) order[dealt(atomicAdd(&hist[mine[i]], 1), B, pair)] = (int32_t)(t + i * 1024);
    for (int64_t b = t + (int64_t)PER_THREAD * 1024; b < B; b += 1024) order[dealt(atomicAdd(&hist[keys[b]], 1), B, pair)] = (int32_t)b;
}

} // namespace dynord

template <typename T>
static void launch_keys(int nf, dim3 grid, hipStream_t st, const T *params, int64_t B, int P, const int32_t *cols, const float *coef,
                        float key_scale, int n_sym, int sym_blocks, int32_t *keys) {
#define DYN_KEYS(NF) hipLaunchKernelGGL((dynord::cost_keys<T, NF>), grid, dim3(256), 0, st, params, B, P, cols, coef, key_scale, n_sym, sym_blocks, keys)
    if (nf <= 4) DYN_KEYS(4);
    else if (nf <= 8) DYN_KEYS(8);
    else if (nf <= 16) DYN_KEYS(16);
    else if (nf <= 24) DYN_KEYS(24);
    else DYN_KEYS(32);
#undef DYN_KEYS
}

extern "C" int32_t dyn_cost_order_capacity(int32_t n_feat) {
    return n_feat <= 4 ? 4 : n_feat <= 8 ? 8 : n_feat <= 16 ? 16 : n_feat <= 24 ? 24 : n_feat <= DYN_MAX_COST_FEATURES ? 32 : -1;
}

extern "C" int dyn_cost_order(const void *params, int32_t dtype, int64_t B, int32_t P, int32_t n_feat, const int32_t *cols,
                              const float *coef, double key_scale, int32_t n_sym, int32_t sym_blocks, int32_t deal_waves_of,
                              int32_t *keys_ws, int32_t *order, void *stream) {
    if (B > 0 && (!params || !cols || !coef || !keys_ws || !order)) return DYN_ERR_NULL;
    if (B < 0 || B > 0x7fffffffLL || P < 1 || n_feat < 1 || n_feat > DYN_MAX_COST_FEATURES) return DYN_ERR_SIZE;
    if (n_feat != dyn_cost_order_capacity(n_feat)) return DYN_ERR_SIZE;   // pad with zero coefficients (dyn_cost_order_capacity)
    if (n_sym < 0 || n_sym > dynord::kMaxSym || sym_blocks < 0 || (n_sym > 1 && (sym_blocks < 2 || sym_blocks * n_sym > P)))
        return DYN_ERR_SIZE;
    if ((dtype != DYN_F32 && dtype != DYN_F64) || !(key_scale > 0.0) || deal_waves_of < 0) return DYN_ERR_OPTS;
    if (B == 0) return 0;
    const dim3 grid((unsigned)((B + 255) / 256));
    if (dtype == DYN_F32)
        launch_keys<float>(n_feat, grid, (hipStream_t)stream, (const float *)params, B, (int)P, cols, coef, (float)key_scale,
                           (int)n_sym, (int)sym_blocks, keys_ws);
    else
        launch_keys<double>(n_feat, grid, (hipStream_t)stream, (const double *)params, B, (int)P, cols, coef, (float)key_scale,
                            (int)n_sym, (int)sym_blocks, keys_ws);
    const int pair = (deal_waves_of > 0 && B % deal_waves_of == 0 && B / deal_waves_of >= 2) ? (int)deal_waves_of : 0;
    hipLaunchKernelGGL(dynord::order_from_keys, dim3(1), dim3(1024), 0, (hipStream_t)stream, keys_ws, B, order, pair);
    return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
}
