// Evaluation metrics on the device (reference: lib/core/evaluate.py:135-166 + lib/utils/eval_utils.py) -- SURVEY.md 8(f) rank 4.
//
//   maed_eval_pose_errors   per frame: visibility mask, pelvis centring (joints 2,3), MPJPE, and the Procrustes-aligned MPJPE
//                           of eval_utils.py:201-252 (batched similarity transform; the reference calls torch.svd on 3x3's)
//   maed_eval_accel         eval_utils.py:10-21 (acceleration) / :24-52 (acceleration error, vis=None)
//   maed_eval_vertex_error  eval_utils.py:88-90 (mean per-vertex distance)
//
// All three are tiny or HBM-bound: one 64-lane wave per frame (lane = joint) for the pose errors so the frame's joints are one
// coalesced read and every reduction is a shuffle tree; one workgroup per frame for the 6890-vertex error.  The 3x3 SVD of the
// Procrustes step is replaced by what the reference actually consumes, R = V Z U^T with Z fixing det(R)=+1: eigenvectors of
// K^T K by cyclic Jacobi (fp64, 3x3), U from K V; building both bases right-handed makes Z the identity for every sign
// convention an SVD routine could pick (see procrustes_rotation), so no branch on det is needed.
#include "common.cuh"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
    // two 32-bit shuffles per step (there is no 64-bit shuffle instruction)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __shfl_xor(lo, o, 64);
        hi = __shfl_xor(hi, o, 64);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

__device__ inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline double normalize3(double* a) {
    const double n = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (n > 0.0) { a[0] /= n; a[1] /= n; a[2] /= n; }
    return n;
}

// R (row-major 3x3) maximising trace(R K) over rotations, K = X1 X2^T = U S V^T  ->  R = V Z U^T (eval_utils.py:226-238).
// With v3 = v1 x v2 and u3 = u1 x u2 both bases have determinant +1; an SVD's own third vectors are +-(these), and the
// reference's Z = diag(1,1,sign det(U V^T)) multiplies exactly that sign back out, so R = V U^T on the right-handed bases.
__device__ inline void procrustes_rotation(const double (&K)[3][3], double (&R)[3][3]) {
    double A[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];   // K^T K
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {       // A <- A J
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {       // A <- J^T A ; V <- V J
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    // the two dominant eigenvectors (columns of V with the largest eigenvalues)
    int i0 = 0;
    if (A[1][1] > A[i0][i0]) i0 = 1;
    if (A[2][2] > A[i0][i0]) i0 = 2;
    int i1 = i0 == 0 ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k != i0 && A[k][k] > A[i1][i1]) i1 = k;
    double v1[3] = {V[0][i0], V[1][i0], V[2][i0]}, v2[3] = {V[0][i1], V[1][i1], V[2][i1]}, v3[3];
    cross3(v1, v2, v3);
    double u1[3], u2[3], u3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u1[r] = K[r][0] * v1[0] + K[r][1] * v1[1] + K[r][2] * v1[2];
        u2[r] = K[r][0] * v2[0] + K[r][1] * v2[1] + K[r][2] * v2[2];
    }
    normalize3(u1);
    const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
    normalize3(u2);
    cross3(u1, u2, u3);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
}

// one wave per frame, lane j = joint j (J <= 64)
// EVAL = true : evaluate.py's protocol (target rows carry visibility; mask, centre on the pelvis, report the two errors)
// EVAL = false: the bare similarity transform of eval_utils.py:201-252 (target rows are x,y,z; report the aligned points)
template <bool EVAL>
__global__ __launch_bounds__(256) void eval_pose_errors_kernel(const float* __restrict__ pred, const float* __restrict__ target, int N, int J,
                                                               float* __restrict__ mpjpe, float* __restrict__ pa_mpjpe,
                                                               float* __restrict__ pred_c, float* __restrict__ target_c,
                                                               float* __restrict__ aligned) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bool on = lane < J;
    float p[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
    if (on) {
        const float* pp = pred + ((int64_t)n * J + lane) * 3;
        const float* gp = target + ((int64_t)n * J + lane) * (EVAL ? 4 : 3);
        const float vis = EVAL ? gp[3] : 1.0f;                     // evaluate.py:142-146: both sides are multiplied by vis
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = pp[c] * vis; g[c] = gp[c] * vis; }
    }
    if constexpr (EVAL) {   // pelvis = midpoint of joints 2 and 3 (evaluate.py:151-155)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float pel_p = 0.5f * (__shfl(p[c], 2, 64) + __shfl(p[c], 3, 64));
            const float pel_g = 0.5f * (__shfl(g[c], 2, 64) + __shfl(g[c], 3, 64));
            p[c] -= pel_p;
            g[c] -= pel_g;
        }
    }
    if (on) {
        if (pred_c) { float* o = pred_c + ((int64_t)n * J + lane) * 3; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
        if (target_c) { float* o = target_c + ((int64_t)n * J + lane) * 3; o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; }
    }
    const float invJ = 1.0f / (float)J;
    const float d0 = p[0] - g[0], d1 = p[1] - g[1], d2 = p[2] - g[2];
    const float e = wave_sum(on ? sqrtf(d0 * d0 + d1 * d1 + d2 * d2) : 0.f) * invJ;       // :158
    // ---- similarity transform S1 = pred -> S2 = target (eval_utils.py:201-252) ----
    double mu1[3], mu2[3], x1[3], x2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mu1[c] = wave_sum_d(on ? (double)p[c] : 0.0) / J;
        mu2[c] = wave_sum_d(on ? (double)g[c] : 0.0) / J;
        x1[c] = on ? (double)p[c] - mu1[c] : 0.0;
        x2[c] = on ? (double)g[c] - mu2[c] : 0.0;
    }
    const double var1 = wave_sum_d(x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2]);
    double K[3][3], R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) K[i][j] = wave_sum_d(x1[i] * x2[j]);
    procrustes_rotation(K, R);                                     // every lane holds the same K: computed redundantly, no broadcast
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) tr += R[i][j] * K[j][i];
    const double scale = tr / var1;                                // var1 == 0 (all joints coincide) is NaN here as in the reference
    float err = 0.f;
    if (on) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double hat = scale * (R[i][0] * x1[0] + R[i][1] * x1[1] + R[i][2] * x1[2]) + mu2[i];   // s R (S1 - mu1) + mu2 == s R S1 + t
            if (aligned) aligned[((int64_t)n * J + lane) * 3 + i] = (float)hat;
            const double d = hat - (double)g[i];
            q += d * d;
        }
        err = (float)sqrt(q);
    }
    const float epa = wave_sum(err) * invJ;                        // :160
    if (lane == 0) {
        if (mpjpe) mpjpe[n] = e;
        if (pa_mpjpe) pa_mpjpe[n] = epa;
    }
}

// out[n] = mean_j || a[n] - 2 a[n+1] + a[n+2] ||, a = joints (- joints_gt when given); one wave per output frame
__global__ __launch_bounds__(256) void eval_accel_kernel(const float* __restrict__ joints, const float* __restrict__ gt, int N, int J, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N - 2) return;
    float acc = 0.f;
    for (int j = lane; j < J; j += 64) {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t i0 = ((int64_t)n * J + j) * 3 + c, st = (int64_t)J * 3;
            float a = joints[i0] - 2.0f * joints[i0 + st] + joints[i0 + 2 * st];
            if (gt) a -= gt[i0] - 2.0f * gt[i0 + st] + gt[i0 + 2 * st];
            q += a * a;
        }
        acc += sqrtf(q);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[n] = acc / (float)J;
}

// out[n] = mean_v || pred[n,v] - target[n,v] ||; one workgroup per frame
__global__ __launch_bounds__(256) void eval_vertex_error_kernel(const float* __restrict__ pred, const float* __restrict__ target, int V, float* __restrict__ out) {
    __shared__ float part[4];
    const int64_t base = (int64_t)blockIdx.x * V * 3;
    float acc = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float* a = pred + base + (int64_t)v * 3;
        const float* b = target + base + (int64_t)v * 3;
        const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
        acc += sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1] + part[2] + part[3]) / (float)V;
}

}  // namespace

extern "C" int maed_eval_pose_errors(const float* pred, const float* target, int N, int J, float* mpjpe, float* pa_mpjpe,
                                     float* pred_centred, float* target_centred, void* stream) {
    MAED_CHECK_ARG(pred && target && mpjpe && pa_mpjpe, MAED_ERR_ARG, "eval_pose_errors: null pointer");
    MAED_CHECK_ARG(N >= 0 && J >= 4 && J <= 64, MAED_ERR_SHAPE, "eval_pose_errors: need 4 <= J <= 64 joints (J=%d), N >= 0", J);
    if (N == 0) return MAED_OK;
    hipLaunchKernelGGL(eval_pose_errors_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, target, N, J, mpjpe, pa_mpjpe,
                       pred_centred, target_centred, (float*)nullptr);
    MAED_CHECK_LAUNCH("eval_pose_errors");
    return MAED_OK;
}

extern "C" int maed_similarity_transform(const float* S1, const float* S2, int N, int J, float* S1_hat, void* stream) {
    MAED_CHECK_ARG(S1 && S2 && S1_hat, MAED_ERR_ARG, "similarity_transform: null pointer");
    MAED_CHECK_ARG(N >= 0 && J >= 1 && J <= 64, MAED_ERR_SHAPE, "similarity_transform: need 1 <= J <= 64 points per set (J=%d), N >= 0", J);
    if (N == 0) return MAED_OK;
    hipLaunchKernelGGL(eval_pose_errors_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, S1, S2, N, J, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, S1_hat);
    MAED_CHECK_LAUNCH("similarity_transform");
    return MAED_OK;
}

extern "C" int maed_eval_accel(const float* joints, const float* joints_gt, int N, int J, float* out, void* stream) {
    MAED_CHECK_ARG(joints && out, MAED_ERR_ARG, "eval_accel: null pointer");
    MAED_CHECK_ARG(N >= 0 && J > 0, MAED_ERR_SHAPE, "eval_accel: bad extents");
    if (N < 3) return MAED_OK;
    hipLaunchKernelGGL(eval_accel_kernel, dim3((N - 2 + 3) / 4), dim3(256), 0, (hipStream_t)stream, joints, joints_gt, N, J, out);
    MAED_CHECK_LAUNCH("eval_accel");
    return MAED_OK;
}

extern "C" int maed_eval_vertex_error(const float* pred_verts, const float* target_verts, int N, int V, float* out, void* stream) {
    MAED_CHECK_ARG(pred_verts && target_verts && out, MAED_ERR_ARG, "eval_vertex_error: null pointer");
    MAED_CHECK_ARG(N >= 0 && V > 0, MAED_ERR_SHAPE, "eval_vertex_error: bad extents");
    if (N == 0) return MAED_OK;
    hipLaunchKernelGGL(eval_vertex_error_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pred_verts, target_verts, V, out);
    MAED_CHECK_LAUNCH("eval_vertex_error");
    return MAED_OK;
}
