// Anomaly-score post-processing on the device: the arithmetic of the reference's
// Predictor.get_score (prediction.py:72-91) and of its threshold evaluation -- find_epsilon, adjust_predicts
// ("point adjust") + calc_point2point for one or many thresholds (epsilon_eval, bf_search; eval_methods.py:6-236).
// O(N) work per threshold on 1-D arrays; the scalar bookkeeping (picking the best z / threshold, F1 from the
// confusion counts) stays on the host in float64 exactly as the reference writes it.
#include "mtadgat_device.h"

namespace mtadgat {

// a[i][d] = |preds - actual| + gamma * |recons - actual|   (the reference writes sqrt((.)**2)); global = mean over d
__global__ void k_eval_scores(const float* __restrict__ preds, const float* __restrict__ recons, const float* __restrict__ actual,
                              long n, int d, long ld_actual, const int* __restrict__ dims, float gamma, float* __restrict__ per_dim,
                              float* __restrict__ global) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int k = 0; k < d; ++k) {
        const float t = actual[i * ld_actual + (dims ? dims[k] : k)];
        const float a = fabsf(preds[i * d + k] - t) + gamma * fabsf(recons[i * d + k] - t);
        if (per_dim) per_dim[i * d + k] = a;
        acc += a;
    }
    if (global) global[i] = acc / (float)d;
}

__device__ __forceinline__ double block_sum(double v, double* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// out[0] += sum e, out[1] += sum e^2  (double)
__global__ void k_eval_moments(const float* __restrict__ e, long n, double* __restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0, s2 = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = e[i];
        s += v;
        s2 += v * v;
    }
    s = block_sum(s, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], s);
        atomicAdd(&out[1], s2);
    }
}

// find_epsilon, one z per blockIdx.y: epsilon = mean + sd * z;
//   out[z] = { sum of e < eps, sum of squares, count of e < eps, |{ i : some |k| <= 49 has e[i + k] >= eps }| }
__global__ void k_eval_epsilon(const float* __restrict__ e, long n, const double* __restrict__ eps, int halo, double* __restrict__ out) {
    __shared__ double sm[256];
    const double ez = eps[blockIdx.y];
    double s = 0.0, s2 = 0.0, cnt = 0.0, dil = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = e[i];
        if (v < ez) { s += v; s2 += v * v; cnt += 1.0; }
        const long lo = i - halo < 0 ? 0 : i - halo, hi = i + halo >= n ? n - 1 : i + halo;
        bool any = false;
        for (long k = lo; k <= hi; ++k) any = any || ((double)e[k] >= ez);
        dil += any ? 1.0 : 0.0;
    }
    s = block_sum(s, sm); s2 = block_sum(s2, sm); cnt = block_sum(cnt, sm); dil = block_sum(dil, sm);
    if (threadIdx.x == 0) {
        double* o = out + 4 * blockIdx.y;
        atomicAdd(&o[0], s); atomicAdd(&o[1], s2); atomicAdd(&o[2], cnt); atomicAdd(&o[3], dil);
    }
}

// runs of label > 0: seg[2 k] = first index, seg[2 k + 1] = last index; *nseg counts them (unordered)
__global__ void k_eval_segments(const unsigned char* __restrict__ label, long n, int* __restrict__ seg, int* __restrict__ nseg, int max_seg) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !label[i] || (i > 0 && label[i - 1])) return;
    long j = i;
    while (j + 1 < n && label[j + 1]) ++j;
    const int k = atomicAdd(nseg, 1);
    if (k < max_seg) { seg[2 * k] = (int)i; seg[2 * k + 1] = (int)j; }
}

// adjust_predicts + calc_point2point for threshold blockIdx.x (eval_methods.py:6-72):
//   predict = score > thr; a detected anomaly segment is predicted in full -- back to its start (the reference's
//   back-fill loop never reaches index 0) and on to its end -- everything else keeps its point-wise prediction.
//   out[t] = { TP, TN, FP, FN, latency sum, detected segments }
__global__ void k_eval_adjust(const float* __restrict__ score, const unsigned char* __restrict__ label, long n, const double* __restrict__ thr,
                              int cmp_f32, const int* __restrict__ seg, const int* __restrict__ nseg_p, double* __restrict__ out) {
    __shared__ double sm[256];
    const double t64 = thr[blockIdx.x];
    const float t32 = (float)t64;
    auto above = [&](float s) { return cmp_f32 ? (s > t32) : ((double)s > t64); };
    double fp = 0.0, neg = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x)
        if (!label[i]) { neg += 1.0; fp += above(score[i]) ? 1.0 : 0.0; }
    double tp = 0.0, fn = 0.0, lat = 0.0, det = 0.0;
    const int nseg = *nseg_p;
    for (int k = threadIdx.x; k < nseg; k += blockDim.x) {
        const int s0 = seg[2 * k], s1 = seg[2 * k + 1];
        int first = -1;
        for (int i = s0; i <= s1; ++i)
            if (above(score[i])) { first = i; break; }
        const double len = (double)(s1 - s0 + 1);
        if (first < 0) {
            fn += len;
        } else {
            det += 1.0;
            const int b0 = s0 > 1 ? s0 : 1;                // back-fill covers j = first .. max(s0, 1)
            if (first > b0) lat += (double)(first - b0);
            if (s0 == 0 && first > 0) { tp += len - 1.0; fn += 1.0; }   // index 0 is never back-filled
            else tp += len;
        }
    }
    fp = block_sum(fp, sm); neg = block_sum(neg, sm); tp = block_sum(tp, sm); fn = block_sum(fn, sm);
    lat = block_sum(lat, sm); det = block_sum(det, sm);
    if (threadIdx.x == 0) {
        double* o = out + 6 * blockIdx.x;
        o[0] = tp; o[1] = neg - fp; o[2] = fp; o[3] = fn; o[4] = lat; o[5] = det;
    }
}

}  // namespace mtadgat

using namespace mtadgat;

extern "C" {

int mtadgat_eval_scores(const float* preds_dev, const float* recons_dev, const float* actual_dev, int64_t n, int d, int64_t ld_actual,
                        const int* dims_dev, float gamma, float* per_dim_dev, float* global_dev, void* stream) {
    if (n <= 0) return 0;
    if (!preds_dev || !recons_dev || !actual_dev || d < 1) return -1;
    hipLaunchKernelGGL(k_eval_scores, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, preds_dev, recons_dev,
                       actual_dev, (long)n, d, (long)ld_actual, dims_dev, gamma, per_dim_dev, global_dev);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// scratch_dev: >= 2 + 4*nz doubles.  out_host: [sum, sumsq] then nz x [pruned sum, pruned sumsq, pruned count, dilated count].
// eps = f(mean, sd) is formed by the caller between the two phases: phase 0 computes the moments, phase 1 the z table.
int mtadgat_eval_moments(const float* e_dev, int64_t n, double* scratch_dev, double* out_host, void* stream) {
    if (!e_dev || !scratch_dev || !out_host || n <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch_dev, 0, 2 * sizeof(double), s) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_eval_moments, dim3(256), dim3(256), 0, s, e_dev, (long)n, scratch_dev);
    if (hipMemcpyAsync(out_host, scratch_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return -3;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -3;
}

int mtadgat_eval_epsilon_table(const float* e_dev, int64_t n, const double* eps_host, int nz, int halo, double* scratch_dev,
                               double* out_host, void* stream) {
    if (!e_dev || !eps_host || !scratch_dev || !out_host || n <= 0 || nz < 1 || nz > 64) return -1;
    hipStream_t s = (hipStream_t)stream;
    double* eps_dev = scratch_dev;               // nz
    double* tab = scratch_dev + 64;              // 4 * nz
    if (hipMemcpyAsync(eps_dev, eps_host, nz * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return -3;
    if (hipMemsetAsync(tab, 0, 4 * nz * sizeof(double), s) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_eval_epsilon, dim3(128, nz), dim3(256), 0, s, e_dev, (long)n, eps_dev, halo, tab);
    if (hipMemcpyAsync(out_host, tab, 4 * nz * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return -3;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -3;
}

// scratch_dev: >= 7*n_thr doubles followed by (2*max_seg + 2) ints;  out_host: n_thr x [TP, TN, FP, FN, latency sum, detected]
int mtadgat_eval_point_adjust(const float* score_dev, const unsigned char* label_dev, int64_t n, const double* thr_host, int n_thr,
                              int compare_f32, int max_seg, double* scratch_dev, double* out_host, void* stream) {
    if (!score_dev || !label_dev || !thr_host || !scratch_dev || !out_host || n <= 0 || n_thr < 1 || max_seg < 1) return -1;
    hipStream_t s = (hipStream_t)stream;
    double* thr_dev = scratch_dev;                                   // n_thr
    double* res = scratch_dev + n_thr;                               // 6 * n_thr
    int* nseg = reinterpret_cast<int*>(scratch_dev + 7 * (size_t)n_thr);
    int* seg = nseg + 2;
    if (hipMemcpyAsync(thr_dev, thr_host, n_thr * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return -3;
    if (hipMemsetAsync(nseg, 0, sizeof(int), s) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_eval_segments, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, label_dev, (long)n, seg, nseg, max_seg);
    int ns = 0;
    if (hipMemcpyAsync(&ns, nseg, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return -3;
    if (hipStreamSynchronize(s) != hipSuccess) return -3;
    if (ns > max_seg) return -5;                                     // more anomaly segments than the scratch holds
    hipLaunchKernelGGL(k_eval_adjust, dim3(n_thr), dim3(256), 0, s, score_dev, label_dev, (long)n, thr_dev, compare_f32, seg, nseg, res);
    if (hipMemcpyAsync(out_host, res, 6 * (size_t)n_thr * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return -3;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -3;
}

}  // extern "C"
