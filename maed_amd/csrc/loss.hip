// lib/core/loss.py LossVideo / LossImage as one fused forward+backward.
//   kp_2d : w * mean(conf * (pred - gt)^2)                                                 loss.py:21-38
//   kp_3d : w * mean(conf * ((pred - pelvis(pred)) - (gt - pelvis(gt)))^2), pelvis = (kp[27] + kp[28]) / 2     :40-62
//   pose  : w * MSE(batch_rodrigues(pred_pose), batch_rodrigues(gt_pose)) over frames with w_smpl           :64-92
//   shape : w * MSE(pred_shape, gt_shape) over the same frames
//   norm  : w * ||pred_theta[:, 3:]||_2 / M3                                                                     :199-201
// Two launches instead of ~90: (1) one wave per frame writes the frame's partial sums and every gradient that needs no
// global quantity; (2) each workgroup re-reduces the partials in a fixed order (deterministic) and finishes d_theta,
// which needs n_valid and the global norm.  Means are taken over the same element counts as the reference's .mean().
#include "common.cuh"
#include "dual.cuh"

// lib/utils/geometry.py:12-56: axis-angle -> quaternion -> rotation matrix (with the reference's 1e-8 offset inside the
// norm and the second normalisation inside quat2mat)
template <typename S>
__device__ __forceinline__ void rodrigues_eval(const S (&a)[3], S (&R)[9]) {
    const S b[3] = {a[0] + 1e-8f, a[1] + 1e-8f, a[2] + 1e-8f};
    const S angle = dsqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    const S half = angle * 0.5f;
    const S c = dcos(half), s = dsin(half);
    S q[4] = {c, s * (a[0] / angle), s * (a[1] / angle), s * (a[2] / angle)};
    const S qn = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const S w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
    const S w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const S wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = xy * 2.0f - wz * 2.0f; R[2] = wy * 2.0f + xz * 2.0f;
    R[3] = wz * 2.0f + xy * 2.0f; R[4] = w2 - x2 + y2 - z2; R[5] = yz * 2.0f - wx * 2.0f;
    R[6] = xz * 2.0f - wy * 2.0f; R[7] = wx * 2.0f + yz * 2.0f; R[8] = w2 - x2 - y2 + z2;
}

enum { P_KP2D = 0, P_KP3D, P_SHAPE, P_POSE, P_NORM2, P_VALID, P_STRIDE = 8 };

__global__ __launch_bounds__(64) void loss_frame_kernel(const float* __restrict__ pred_kp2d, const float* __restrict__ gt_kp2d, int M2,
                                                        const float* __restrict__ pred_kp3d, const float* __restrict__ gt_kp3d,
                                                        const float* __restrict__ pred_theta, const float* __restrict__ gt_theta,
                                                        const uint8_t* __restrict__ w_smpl, int M3, maed_loss_weights wt,
                                                        float* __restrict__ d_kp2d, float* __restrict__ d_kp3d, float* __restrict__ d_theta,
                                                        double* __restrict__ partials) {
    const int64_t m = blockIdx.x;
    const int lane = threadIdx.x;
    float s2d = 0.f, s3d = 0.f, sshape = 0.f, spose = 0.f, snorm = 0.f;
    // ---- 2D keypoints
    if (m < M2 && lane < 49) {
        const float* p = pred_kp2d + (m * 49 + lane) * 2;
        const float* g = gt_kp2d + (m * 49 + lane) * 3;
        const float conf = g[2], e0 = p[0] - g[0], e1 = p[1] - g[1];
        s2d = conf * (e0 * e0 + e1 * e1);
        const float k = wt.w_kp2d * 2.0f * conf / ((float)M2 * 98.0f);
        d_kp2d[(m * 49 + lane) * 2 + 0] = k * e0;
        d_kp2d[(m * 49 + lane) * 2 + 1] = k * e1;
    }
    s2d = wave_sum(s2d);
    bool valid = false;
    if (m < M3) {
        // ---- 3D keypoints, pelvis-centred
        float e[3] = {0.f, 0.f, 0.f};
        float conf = 0.f;
        if (gt_kp3d) {
            const float* P = pred_kp3d + m * 49 * 3;
            const float* G = gt_kp3d + m * 49 * 4;
            if (lane < 49) {
                conf = G[lane * 4 + 3];
                for (int c = 0; c < 3; ++c) {
                    const float pp = 0.5f * (P[27 * 3 + c] + P[28 * 3 + c]), gp = 0.5f * (G[27 * 4 + c] + G[28 * 4 + c]);
                    e[c] = (P[lane * 3 + c] - pp) - (G[lane * 4 + c] - gp);
                }
                s3d = conf * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            }
            const float k = wt.w_kp3d * 2.0f / ((float)M3 * 147.0f);
            float ge[3];
            for (int c = 0; c < 3; ++c) ge[c] = lane < 49 ? k * conf * e[c] : 0.f;
            float tot[3];
            for (int c = 0; c < 3; ++c) tot[c] = wave_sum(ge[c]);     // -> -d pelvis
            if (lane < 49)
                for (int c = 0; c < 3; ++c)
                    d_kp3d[(m * 49 + lane) * 3 + c] = ge[c] - ((lane == 27 || lane == 28) ? 0.5f * tot[c] : 0.f);
        } else if (lane < 49) {
            for (int c = 0; c < 3; ++c) d_kp3d[(m * 49 + lane) * 3 + c] = 0.f;
        }
        s3d = wave_sum(s3d);
        // ---- SMPL parameters
        const float* th = pred_theta + m * 85;
        const float* gt = gt_theta + m * 85;
        valid = w_smpl[m] != 0;
        float* dth = d_theta + m * 85;
        if (lane < 3) dth[lane] = 0.f;                                   // the camera carries no loss term
        if (lane < 24) {
            float gr[3] = {0.f, 0.f, 0.f};
            if (valid) {
                typedef Dual<3> D;
                D a[3], R[9];
                float ag[3], Rg[9];
                for (int c = 0; c < 3; ++c) { a[c] = seed<3>(th[3 + lane * 3 + c], c); ag[c] = gt[3 + lane * 3 + c]; }
                rodrigues_eval<D>(a, R);
                rodrigues_eval<float>(ag, Rg);
                for (int k = 0; k < 9; ++k) {
                    const float diff = R[k].v - Rg[k];
                    spose += diff * diff;
                    for (int c = 0; c < 3; ++c) gr[c] = fmaf(2.0f * diff, R[k].d[c], gr[c]);
                }
            }
            for (int c = 0; c < 3; ++c) dth[3 + lane * 3 + c] = gr[c];   // raw: scaled by w_pose / (n_valid * 216) in pass 2
        }
        if (lane < 10) {
            const float d = th[75 + lane] - gt[75 + lane];
            if (valid) sshape = d * d;
            dth[75 + lane] = valid ? 2.0f * d : 0.f;                     // raw: scaled by w_shape / (n_valid * 10) in pass 2
        }
        for (int i = lane; i < 82; i += 64) snorm += th[3 + i] * th[3 + i];
        spose = wave_sum(spose); sshape = wave_sum(sshape); snorm = wave_sum(snorm);
    }
    if (lane == 0) {
        double* o = partials + m * P_STRIDE;
        o[P_KP2D] = s2d; o[P_KP3D] = s3d; o[P_SHAPE] = sshape; o[P_POSE] = spose; o[P_NORM2] = snorm; o[P_VALID] = valid ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ pred_theta, int M2, int M3, int has3d, maed_loss_weights wt,
                                                          const double* __restrict__ partials, float* __restrict__ losses,
                                                          float* __restrict__ d_theta) {
    __shared__ double s_part[256][6];
    __shared__ double s_tot[6];
    const int M = M2 > M3 ? M2 : M3;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int m = threadIdx.x; m < M; m += 256)
        for (int k = 0; k < 6; ++k) acc[k] += partials[(int64_t)m * P_STRIDE + k];
    for (int k = 0; k < 6; ++k) s_part[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        double s = 0;
        for (int t = 0; t < 256; ++t) s += s_part[t][threadIdx.x];
        s_tot[threadIdx.x] = s;
    }
    __syncthreads();
    const double nv = s_tot[P_VALID];
    const double norm = sqrt(s_tot[P_NORM2]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double l2d = M2 > 0 ? wt.w_kp2d * s_tot[P_KP2D] / ((double)M2 * 98.0) : 0.0;
        const double l3d = (M3 > 0 && has3d) ? wt.w_kp3d * s_tot[P_KP3D] / ((double)M3 * 147.0) : 0.0;
        const double lsh = nv > 0 ? wt.w_shape * s_tot[P_SHAPE] / (nv * 10.0) : 0.0;
        const double lps = nv > 0 ? wt.w_pose * s_tot[P_POSE] / (nv * 216.0) : 0.0;
        const double lnm = M3 > 0 ? wt.w_norm * norm / (double)M3 : 0.0;
        losses[0] = (float)l2d; losses[1] = (float)l3d; losses[2] = (float)lsh; losses[3] = (float)lps; losses[4] = (float)lnm;
        losses[5] = (float)(l2d + l3d + lsh + lps + lnm); losses[6] = (float)nv; losses[7] = 0.f;
    }
    const float kpose = nv > 0 ? (float)(wt.w_pose / (nv * 216.0)) : 0.f;
    const float kshape = nv > 0 ? (float)(wt.w_shape / (nv * 10.0)) : 0.f;
    const float knorm = (M3 > 0 && norm > 0) ? (float)(wt.w_norm / (norm * (double)M3)) : 0.f;
    const int64_t n = (int64_t)M3 * 82;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / 82; const int c = 3 + (int)(i % 82);
        const int64_t o = m * 85 + c;
        d_theta[o] = d_theta[o] * (c < 75 ? kpose : kshape) + knorm * pred_theta[o];
    }
}

extern "C" int maed_loss_fwd_bwd(const float* pred_kp2d, const float* gt_kp2d, int M2, const float* pred_kp3d, const float* gt_kp3d,
                                 const float* pred_theta, const float* gt_theta, const uint8_t* w_smpl, int M3, const maed_loss_weights* w,
                                 float* losses, float* d_kp2d, float* d_kp3d, float* d_theta, double* partials, void* stream) {
    MAED_CHECK_ARG(w && losses && partials, MAED_ERR_ARG, "loss_fwd_bwd: null pointer");
    MAED_CHECK_ARG(M2 >= 0 && M3 >= 0, MAED_ERR_SHAPE, "loss_fwd_bwd: negative frame count");
    MAED_CHECK_ARG(M2 == 0 || (pred_kp2d && gt_kp2d && d_kp2d), MAED_ERR_ARG, "loss_fwd_bwd: null 2D keypoint pointer");
    MAED_CHECK_ARG(M3 == 0 || (pred_kp3d && pred_theta && gt_theta && w_smpl && d_kp3d && d_theta), MAED_ERR_ARG, "loss_fwd_bwd: null 3D/SMPL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int M = M2 > M3 ? M2 : M3;
    if (M > 0)
        hipLaunchKernelGGL(loss_frame_kernel, dim3(M), dim3(64), 0, s, pred_kp2d, gt_kp2d, M2, pred_kp3d, gt_kp3d, pred_theta, gt_theta, w_smpl, M3,
                           *w, d_kp2d, d_kp3d, d_theta, partials);
    const int blocks = M3 > 0 ? (int)(((int64_t)M3 * 82 + 255) / 256) : 1;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(blocks > 64 ? 64 : blocks), dim3(256), 0, s, pred_theta, M2, M3, gt_kp3d ? 1 : 0, *w, partials, losses,
                       d_theta);
    MAED_CHECK_LAUNCH("loss_fwd_bwd");
    return MAED_OK;
}


// ---- acceleration term (loss.py:94-117, weight e_smpl_accl_loss; off in the shipped configs) ------------------------------------------
//   L = w * mean_{n, t < T-2, j, c} ( conf[n][t+2][j]^4 * ((p[t+2] - 2 p[t+1] + p[t]) - (g[t+2] - 2 g[t+1] + g[t])) )^2
// (conf_velocity = conf[1:]^2, conf_accl = conf_velocity[1:]^2 -- the reference squares the confidence twice).  One thread per (clip,
// frame s, joint): it owns the value terms of t = s and gathers the gradient of p[s] from the <= 3 second differences that contain it.
__global__ __launch_bounds__(256) void loss_accl_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int N, int T, float w,
                                                        float* __restrict__ d_pred, double* __restrict__ loss) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * T * 49;
    float val = 0.f;
    if (i < total) {
        const int j = (int)(i % 49), s = (int)((i / 49) % T);
        const int64_t n = i / (49 * (int64_t)T);
        const float scale = 2.0f * w / ((float)N * (float)(T - 2) * 147.0f);
        float g3[3] = {0.f, 0.f, 0.f};
        for (int t = s - 2; t <= s; ++t) {
            if (t < 0 || t > T - 3) continue;
            const int64_t b0 = ((n * T + t) * 49 + j), b1 = b0 + 49, b2 = b1 + 49;
            const float c = gt[b2 * 4 + 3], c2 = c * c, c4 = c2 * c2, c8 = c4 * c4;
            const float coef = (t == s - 1) ? -2.0f : 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = (pred[b2 * 3 + k] - 2.0f * pred[b1 * 3 + k] + pred[b0 * 3 + k]) - (gt[b2 * 4 + k] - 2.0f * gt[b1 * 4 + k] + gt[b0 * 4 + k]);
                g3[k] += coef * c8 * e;
                if (t == s) val += c8 * e * e;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) d_pred[i * 3 + k] = scale * g3[k];
    }
    val = wave_sum(val);
    if ((threadIdx.x & 63) == 0 && val != 0.f) atomicAdd(loss, (double)val * (double)w / ((double)N * (double)(T - 2) * 147.0));
}

extern "C" int maed_loss_accl_fwd_bwd(const float* pred_kp3d, const float* gt_kp3d, int N, int T, float weight, double* loss, float* d_kp3d, void* stream) {
    MAED_CHECK_ARG(pred_kp3d && gt_kp3d && loss && d_kp3d, MAED_ERR_ARG, "loss_accl_fwd_bwd: null pointer");
    MAED_CHECK_ARG(N >= 0 && T >= 3, MAED_ERR_SHAPE, "loss_accl_fwd_bwd: needs clips of at least 3 frames (T=%d)", T);
    MAED_HIP(hipMemsetAsync(loss, 0, sizeof(double), (hipStream_t)stream), "loss: memset");
    if (N == 0) return MAED_OK;
    const int64_t total = (int64_t)N * T * 49;
    hipLaunchKernelGGL(loss_accl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_kp3d, gt_kp3d, N, T, weight, d_kp3d, loss);
    MAED_CHECK_LAUNCH("loss_accl_fwd_bwd");
    return MAED_OK;
}
