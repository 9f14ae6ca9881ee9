// Decoder tail BACKWARD (training graph of lib/models/ktd.py:69-124): projection + integer joint gather (K14/K15),
// SMPL linear blend skinning (K12: vertex part and kinematic chain), 6D -> rotmat -> axis-angle (K11) and the KTD
// ancestor chain (K10), plus the pack/unpack of the 26 regressor weights into the operands of one head GEMM.
// ~10 launches replace the ~1000 tiny ATen kernels autograd needs for the same graph.  fp32 throughout; these are
// (F, .)-row problems, HBM/latency bound: thread-per-frame / thread-per-vertex kernels, no MFMA.
#include "common.cuh"
#include "ktd_tables.cuh"
#include "dual.cuh"

// ---- K14/K15 backward ---------------------------------------------------------------------------------------------------
// one wave per frame: lane jj < 49 differentiates the projection of joint jj (spin.py:113-157), the wave reduces the
// camera gradient, then lane idx < 54 gathers the scatter-add through joint_map (smpl.py:98-99) in index order.
__global__ __launch_bounds__(64) void joints_project_bwd_kernel(const float* __restrict__ kp3d, const float* __restrict__ cam,
                                                                const int64_t* __restrict__ joint_map, const float* __restrict__ d_kp3d,
                                                                const float* __restrict__ d_kp2d, const float* __restrict__ d_cam_in,
                                                                int64_t cam_in_stride, float* __restrict__ d_j24, float* __restrict__ d_e21,
                                                                float* __restrict__ d_e9, float* __restrict__ d_cam, int F) {
    __shared__ float s_dp[49][3];
    __shared__ int s_map[49];
    const int64_t f = blockIdx.x;
    const int lane = threadIdx.x;
    const float* cm = cam + f * 3;
    const float den = 224.0f * cm[0] + 1e-9f;
    float c1 = 0.f, c2 = 0.f, ctz = 0.f;
    if (lane < 49) {
        const int64_t i = f * 49 + lane;
        const float tz = 2.0f * 5000.0f / den;
        const float X = kp3d[i * 3 + 0] + cm[1], Y = kp3d[i * 3 + 1] + cm[2], Z = kp3d[i * 3 + 2] + tz;
        const float s = 5000.0f / 112.0f;
        const float du = d_kp2d ? d_kp2d[i * 2 + 0] : 0.f, dv = d_kp2d ? d_kp2d[i * 2 + 1] : 0.f;
        c1 = s * du / Z; c2 = s * dv / Z; ctz = -s * (du * X + dv * Y) / (Z * Z);
        s_dp[lane][0] = c1 + (d_kp3d ? d_kp3d[i * 3 + 0] : 0.f);
        s_dp[lane][1] = c2 + (d_kp3d ? d_kp3d[i * 3 + 1] : 0.f);
        s_dp[lane][2] = ctz + (d_kp3d ? d_kp3d[i * 3 + 2] : 0.f);
        s_map[lane] = (int)joint_map[lane];
    }
    c1 = wave_sum(c1); c2 = wave_sum(c2); ctz = wave_sum(ctz);
    __syncthreads();
    if (lane == 0) {
        const float* ci = d_cam_in ? d_cam_in + f * cam_in_stride : nullptr;
        d_cam[f * 3 + 0] = ctz * (-2.0f * 5000.0f * 224.0f / (den * den)) + (ci ? ci[0] : 0.f);
        d_cam[f * 3 + 1] = c1 + (ci ? ci[1] : 0.f);
        d_cam[f * 3 + 2] = c2 + (ci ? ci[2] : 0.f);
    }
    if (lane < 54) {
        float g[3] = {0.f, 0.f, 0.f};
        for (int jj = 0; jj < 49; ++jj)
            if (s_map[jj] == lane) { g[0] += s_dp[jj][0]; g[1] += s_dp[jj][1]; g[2] += s_dp[jj][2]; }
        float* dst = lane < 24 ? d_j24 + (f * 24 + lane) * 3 : lane < 45 ? d_e21 + (f * 21 + (lane - 24)) * 3 : d_e9 + (f * 9 + (lane - 45)) * 3;
        dst[0] = g[0]; dst[1] = g[1]; dst[2] = g[2];
    }
}

extern "C" int maed_smpl_joints_project_bwd(const float* kp3d, const float* cam, const int64_t* joint_map, const float* d_kp3d,
                                            const float* d_kp2d, const float* d_cam_in, int64_t cam_in_stride, float* d_joints24,
                                            float* d_extra21, float* d_extra9, float* d_cam, int F, void* stream) {
    MAED_CHECK_ARG(kp3d && cam && joint_map && d_joints24 && d_extra21 && d_extra9 && d_cam, MAED_ERR_ARG, "smpl_joints_project_bwd: null pointer");
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(joints_project_bwd_kernel, dim3(F), dim3(64), 0, (hipStream_t)stream, kp3d, cam, joint_map, d_kp3d, d_kp2d, d_cam_in,
                       cam_in_stride, d_joints24, d_extra21, d_extra9, d_cam, F);
    MAED_CHECK_LAUNCH("smpl_joints_project_bwd");
    return MAED_OK;
}

// ---- K12 backward, vertex part ------------------------------------------------------------------------------------------
// thread per vertex, SK_FB frames per workgroup (lbs_weights / Jextra columns are read once per SK_FB frames).
// verts_v = T_v [vp_v, 1],  T_v = sum_j w_vj A_j  =>  d vp_v = T_v[:, :3]^T d_v,   dA_j += w_vj d_v [vp_v, 1]^T.
// dA is reduced in LDS (ds_add_f32; only vertices with a non-zero d_v and joints with a non-zero weight take part --
// with the 9+21 extra joints as the only consumers that is ~300 of 6890 vertices) and flushed with one global atomic
// per (frame, joint, entry) per workgroup.
#define SK_FB 4
__global__ __launch_bounds__(256) void smpl_skin_bwd_kernel(maed_smpl_params sp, const float* __restrict__ A, const float* __restrict__ v_posed,
                                                            const float* __restrict__ d_verts, const float* __restrict__ d_e21,
                                                            const int64_t* __restrict__ extra_ids, const float* __restrict__ d_e9,
                                                            const float* __restrict__ Jextra, float* __restrict__ d_vposed,
                                                            float* __restrict__ dA, int F) {
    __shared__ float s_A[SK_FB][NJ * 12];
    __shared__ float s_dA[SK_FB][NJ * 12];
    __shared__ float s_de9[SK_FB][27];
    __shared__ float s_de21[SK_FB][63];
    __shared__ int s_ids[21];
    const int f0 = blockIdx.y * SK_FB;
    for (int i = threadIdx.x; i < SK_FB * NJ * 12; i += 256) {
        const int fb = i / (NJ * 12), k = i % (NJ * 12);
        s_A[fb][k] = A[(int64_t)min(f0 + fb, F - 1) * NJ * 12 + k];
        s_dA[fb][k] = 0.f;
    }
    for (int i = threadIdx.x; i < SK_FB * 27; i += 256) s_de9[i / 27][i % 27] = d_e9[(int64_t)min(f0 + i / 27, F - 1) * 27 + i % 27];
    for (int i = threadIdx.x; i < SK_FB * 63; i += 256) s_de21[i / 63][i % 63] = d_e21[(int64_t)min(f0 + i / 63, F - 1) * 63 + i % 63];
    if (threadIdx.x < 21) s_ids[threadIdx.x] = (int)extra_ids[threadIdx.x];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < NV) {
        float w[NJ], jx[9];
        for (int j = 0; j < NJ; ++j) w[j] = sp.lbs_weights[v * NJ + j];
        for (int r = 0; r < 9; ++r) jx[r] = Jextra[(int64_t)r * NV + v];
#pragma unroll
        for (int fb = 0; fb < SK_FB; ++fb) {
            const int f = f0 + fb;
            if (f >= F) break;
            const int64_t o = ((int64_t)f * NV + v) * 3;
            float dv[3] = {0.f, 0.f, 0.f};
            if (d_verts) { dv[0] = d_verts[o]; dv[1] = d_verts[o + 1]; dv[2] = d_verts[o + 2]; }
            for (int r = 0; r < 9; ++r)
                for (int c = 0; c < 3; ++c) dv[c] = fmaf(jx[r], s_de9[fb][r * 3 + c], dv[c]);
            for (int k = 0; k < 21; ++k)
                if (s_ids[k] == v) { dv[0] += s_de21[fb][k * 3]; dv[1] += s_de21[fb][k * 3 + 1]; dv[2] += s_de21[fb][k * 3 + 2]; }
            float T[12];
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
            for (int j = 0; j < NJ; ++j)
                for (int e = 0; e < 12; ++e) T[e] = fmaf(w[j], s_A[fb][j * 12 + e], T[e]);
            d_vposed[o + 0] = T[0] * dv[0] + T[4] * dv[1] + T[8] * dv[2];
            d_vposed[o + 1] = T[1] * dv[0] + T[5] * dv[1] + T[9] * dv[2];
            d_vposed[o + 2] = T[2] * dv[0] + T[6] * dv[1] + T[10] * dv[2];
            if (dv[0] != 0.f || dv[1] != 0.f || dv[2] != 0.f) {
                const float vh[4] = {v_posed[o], v_posed[o + 1], v_posed[o + 2], 1.0f};
                for (int j = 0; j < NJ; ++j) {
                    if (w[j] == 0.f) continue;
                    for (int r = 0; r < 3; ++r)
                        for (int e = 0; e < 4; ++e) atomicAdd(&s_dA[fb][j * 12 + r * 4 + e], w[j] * dv[r] * vh[e]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SK_FB * NJ * 12; i += 256) {
        const int fb = i / (NJ * 12), k = i % (NJ * 12);
        const float x = s_dA[fb][k];
        if (f0 + fb < F && x != 0.f) atomicAdd(dA + (int64_t)(f0 + fb) * NJ * 12 + k, x);
    }
}

extern "C" int maed_smpl_skin_bwd(const maed_smpl_params* sp, const float* A, const float* v_posed, const float* d_verts,
                                  const float* d_extra21, const int64_t* extra_vertex_ids, const float* d_extra9, const float* Jextra,
                                  float* d_vposed, float* dA, int F, void* stream) {
    MAED_CHECK_ARG(sp && A && v_posed && d_extra21 && extra_vertex_ids && d_extra9 && Jextra && d_vposed && dA, MAED_ERR_ARG, "smpl_skin_bwd: null pointer");
    MAED_CHECK_ARG(sp->lbs_weights, MAED_ERR_ARG, "smpl_skin_bwd: null SMPL parameter");
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(smpl_skin_bwd_kernel, dim3((NV + 255) / 256, (F + SK_FB - 1) / SK_FB), dim3(256), 0, (hipStream_t)stream, *sp, A, v_posed,
                       d_verts, d_extra21, extra_vertex_ids, d_extra9, Jextra, d_vposed, dA, F);
    MAED_CHECK_LAUNCH("smpl_skin_bwd");
    return MAED_OK;
}

// ---- K12 backward, vertex part, ACTIVE vertices only ---------------------------------------------------------------------------
// Without a gradient on the vertices themselves (the training objective of lib/core/loss.py reads key points only) d_v is non-zero exactly on the vertices the
// 9 regressed + 21 selected extra joints read: the non-zero columns of J_regressor_extra and extra_vertex_ids -- ~290 of 6890.  The dense kernel above spends
// 24 x 12 FMAs per (vertex, frame) on transforms that then multiply zeros, and the pose-corrective GEMM behind it contracts 20670 columns of which ~870 are
// non-zero.  Here: thread per (active vertex, frame), compact d_vposed (F, na, 3); the GEMM runs on the matching columns of [posedirs; shapedirs^T]
// (SMPL.pose_shape_dirs_active).  Same arithmetic per active vertex, same LDS reduction of dA.
#define SKS_VB 64
__global__ __launch_bounds__(256) void smpl_skin_bwd_sparse_kernel(maed_smpl_params sp, const float* __restrict__ A, const float* __restrict__ v_posed,
                                                                   const int* __restrict__ active, int na, const float* __restrict__ d_e21,
                                                                   const int64_t* __restrict__ extra_ids, const float* __restrict__ d_e9,
                                                                   const float* __restrict__ Jextra, float* __restrict__ d_vposed_a, float* __restrict__ dA, int F) {
    __shared__ float s_A[SK_FB][NJ * 12];
    __shared__ float s_dA[SK_FB][NJ * 12];
    __shared__ float s_de9[SK_FB][27];
    __shared__ float s_de21[SK_FB][63];
    __shared__ int s_ids[21];
    const int f0 = blockIdx.y * SK_FB;
    for (int i = threadIdx.x; i < SK_FB * NJ * 12; i += 256) {
        const int fb = i / (NJ * 12), k = i % (NJ * 12);
        s_A[fb][k] = A[(int64_t)min(f0 + fb, F - 1) * NJ * 12 + k];
        s_dA[fb][k] = 0.f;
    }
    for (int i = threadIdx.x; i < SK_FB * 27; i += 256) s_de9[i / 27][i % 27] = d_e9[(int64_t)min(f0 + i / 27, F - 1) * 27 + i % 27];
    for (int i = threadIdx.x; i < SK_FB * 63; i += 256) s_de21[i / 63][i % 63] = d_e21[(int64_t)min(f0 + i / 63, F - 1) * 63 + i % 63];
    if (threadIdx.x < 21) s_ids[threadIdx.x] = (int)extra_ids[threadIdx.x];
    __syncthreads();
    const int a = blockIdx.x * SKS_VB + (threadIdx.x & (SKS_VB - 1)), fb = threadIdx.x / SKS_VB, f = f0 + fb;
    if (a < na && f < F) {
        const int v = active[a];
        float dv[3] = {0.f, 0.f, 0.f};
        for (int r = 0; r < 9; ++r) {
            const float jx = Jextra[(int64_t)r * NV + v];
            for (int c = 0; c < 3; ++c) dv[c] = fmaf(jx, s_de9[fb][r * 3 + c], dv[c]);
        }
        for (int k = 0; k < 21; ++k)
            if (s_ids[k] == v) { dv[0] += s_de21[fb][k * 3]; dv[1] += s_de21[fb][k * 3 + 1]; dv[2] += s_de21[fb][k * 3 + 2]; }
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float w = sp.lbs_weights[v * NJ + j];
            for (int e = 0; e < 12; ++e) T[e] = fmaf(w, s_A[fb][j * 12 + e], T[e]);
        }
        float* q = d_vposed_a + ((int64_t)f * na + a) * 3;
        q[0] = T[0] * dv[0] + T[4] * dv[1] + T[8] * dv[2];
        q[1] = T[1] * dv[0] + T[5] * dv[1] + T[9] * dv[2];
        q[2] = T[2] * dv[0] + T[6] * dv[1] + T[10] * dv[2];
        if (dv[0] != 0.f || dv[1] != 0.f || dv[2] != 0.f) {
            const int64_t o = ((int64_t)f * NV + v) * 3;
            const float vh[4] = {v_posed[o], v_posed[o + 1], v_posed[o + 2], 1.0f};
            for (int j = 0; j < NJ; ++j) {
                const float w = sp.lbs_weights[v * NJ + j];
                if (w == 0.f) continue;
                for (int r = 0; r < 3; ++r)
                    for (int e = 0; e < 4; ++e) atomicAdd(&s_dA[fb][j * 12 + r * 4 + e], w * dv[r] * vh[e]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SK_FB * NJ * 12; i += 256) {
        const int fb2 = i / (NJ * 12), k = i % (NJ * 12);
        const float x = s_dA[fb2][k];
        if (f0 + fb2 < F && x != 0.f) atomicAdd(dA + (int64_t)(f0 + fb2) * NJ * 12 + k, x);
    }
}

extern "C" int maed_smpl_skin_bwd_sparse(const maed_smpl_params* sp, const float* A, const float* v_posed, const int32_t* active, int n_active,
                                         const float* d_extra21, const int64_t* extra_vertex_ids, const float* d_extra9, const float* Jextra,
                                         float* d_vposed_active, float* dA, int F, void* stream) {
    MAED_CHECK_ARG(sp && A && v_posed && active && d_extra21 && extra_vertex_ids && d_extra9 && Jextra && d_vposed_active && dA, MAED_ERR_ARG, "smpl_skin_bwd_sparse: null pointer");
    MAED_CHECK_ARG(sp->lbs_weights, MAED_ERR_ARG, "smpl_skin_bwd_sparse: null SMPL parameter");
    MAED_CHECK_ARG(n_active > 0 && n_active <= NV, MAED_ERR_SHAPE, "smpl_skin_bwd_sparse: n_active=%d out of range", n_active);
    static_assert(SKS_VB * SK_FB == 256, "one thread per (active vertex, frame) of the block");
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(smpl_skin_bwd_sparse_kernel, dim3((n_active + SKS_VB - 1) / SKS_VB, (F + SK_FB - 1) / SK_FB), dim3(256), 0, (hipStream_t)stream, *sp, A, v_posed,
                       (const int*)active, n_active, d_extra21, extra_vertex_ids, d_extra9, Jextra, d_vposed_active, dA, F);
    MAED_CHECK_LAUNCH("smpl_skin_bwd_sparse");
    return MAED_OK;
}

// ---- K12 backward, kinematic chain -----------------------------------------------------------------------------------------
// recompute J, Rw, tw, then walk the tree from the leaves.
//   forward:  Rw_j = Rw_p R_j,  tw_j = Rw_p (J_j - J_p) + tw_p,  A_j = [Rw_j | tw_j - Rw_j J_j],  joints24_j = tw_j
// 32 lanes per frame, 2 frames per workgroup, all chain state in LDS (2.9 KB per frame).  Joints are visited one after another where the
// recurrence requires it (children accumulate into their parent in a fixed order); what runs in parallel are the independent outputs of each step.
#define SC_FPB 2
__global__ __launch_bounds__(64) void smpl_chain_bwd_par_kernel(maed_smpl_params sp, const float* __restrict__ betas, const float* __restrict__ rotmat,
                                                                const float* __restrict__ dA, const float* __restrict__ d_j24,
                                                                const float* __restrict__ dpf_dbeta, const float* __restrict__ d_rot_in,
                                                                const float* __restrict__ d_betas_in, int64_t betas_in_stride,
                                                                float* __restrict__ d_rotmat, float* __restrict__ d_betas, int F) {
    __shared__ float J[SC_FPB][NJ][3], Rw[SC_FPB][NJ][9], tw[SC_FPB][NJ][3], gRw[SC_FPB][NJ][9], gtw[SC_FPB][NJ][3], gJ[SC_FPB][NJ][3];
    const int l = threadIdx.x & 31, fs = threadIdx.x >> 5;
    const int f = blockIdx.x * SC_FPB + fs, fc = f < F ? f : F - 1;
    const float* b = betas + (int64_t)fc * 10;
    for (int t = l; t < NJ * 3; t += 32) {
        float s = sp.J_template[t];
        for (int k = 0; k < 10; ++k) s = fmaf(sp.J_shapedirs[t * 10 + k], b[k], s);
        J[fs][t / 3][t % 3] = s;
    }
    __syncthreads();
    const float* R = rotmat + (int64_t)fc * NJ * 9;
    for (int j = 0; j < NJ; ++j) {                               // forward chain, 12 outputs per joint
        const int p = sp.parents[j];
        const float* Rj = R + j * 9;
        if (l < 12) {
            if (p < 0) {
                if (l < 9) Rw[fs][j][l] = Rj[l];
                else tw[fs][j][l - 9] = J[fs][j][l - 9];
            } else if (l < 9) {
                const int r = l / 3, c = l % 3;
                Rw[fs][j][l] = Rw[fs][p][r * 3 + 0] * Rj[0 * 3 + c] + Rw[fs][p][r * 3 + 1] * Rj[1 * 3 + c] + Rw[fs][p][r * 3 + 2] * Rj[2 * 3 + c];
            } else {
                const int r = l - 9;
                const float rel[3] = {J[fs][j][0] - J[fs][p][0], J[fs][j][1] - J[fs][p][1], J[fs][j][2] - J[fs][p][2]};
                tw[fs][j][r] = Rw[fs][p][r * 3 + 0] * rel[0] + Rw[fs][p][r * 3 + 1] * rel[1] + Rw[fs][p][r * 3 + 2] * rel[2] + tw[fs][p][r];
            }
        }
        __syncthreads();
    }
    if (l < NJ) {                                                // outputs -> chain variables: one joint per lane, serial code per joint
        const int j = l;
        const float* g = dA + ((int64_t)fc * NJ + j) * 12;
        const float* gj = d_j24 + ((int64_t)fc * NJ + j) * 3;
        for (int c = 0; c < 3; ++c) gJ[fs][j][c] = 0.f;
        for (int r = 0; r < 3; ++r) {
            const float gt = g[r * 4 + 3];
            for (int c = 0; c < 3; ++c) {
                gRw[fs][j][r * 3 + c] = g[r * 4 + c] - gt * J[fs][j][c];
                gJ[fs][j][c] -= Rw[fs][j][r * 3 + c] * gt;
            }
            gtw[fs][j][r] = gt + gj[r];
        }
    }
    __syncthreads();
    float* gR = d_rotmat + (int64_t)fc * NJ * 9;
    const float* gin = d_rot_in ? d_rot_in + (int64_t)fc * NJ * 9 : nullptr;
    const float* gpf = dpf_dbeta + (int64_t)fc * 217;
    const bool live = f < F;
    for (int j = NJ - 1; j >= 1; --j) {                          // leaves -> root; lanes 0-8 dR_j, 9-17 gRw[p], 18-20 gJ, 21-23 gtw[p]
        const int p = sp.parents[j];
        const float* Rj = R + j * 9;
        if (l < 9) {
            const int a = l / 3, c = l % 3;
            const float s = Rw[fs][p][0 * 3 + a] * gRw[fs][j][0 * 3 + c] + Rw[fs][p][1 * 3 + a] * gRw[fs][j][1 * 3 + c] + Rw[fs][p][2 * 3 + a] * gRw[fs][j][2 * 3 + c];
            if (live) gR[j * 9 + a * 3 + c] = s + gpf[(j - 1) * 9 + a * 3 + c] + (gin ? gin[j * 9 + a * 3 + c] : 0.f);
        } else if (l < 18) {
            const int r = (l - 9) / 3, a = (l - 9) % 3;
            const float rel_a = J[fs][j][a] - J[fs][p][a];
            gRw[fs][p][r * 3 + a] += gRw[fs][j][r * 3 + 0] * Rj[a * 3 + 0] + gRw[fs][j][r * 3 + 1] * Rj[a * 3 + 1] + gRw[fs][j][r * 3 + 2] * Rj[a * 3 + 2]
                                     + gtw[fs][j][r] * rel_a;
        } else if (l < 21) {
            const int a = l - 18;
            const float grel = Rw[fs][p][0 * 3 + a] * gtw[fs][j][0] + Rw[fs][p][1 * 3 + a] * gtw[fs][j][1] + Rw[fs][p][2 * 3 + a] * gtw[fs][j][2];
            gJ[fs][j][a] += grel; gJ[fs][p][a] -= grel;
        } else if (l < 24) {
            const int r = l - 21;
            gtw[fs][p][r] += gtw[fs][j][r];
        }
        __syncthreads();
    }
    if (live && l < 9) gR[l] = gRw[fs][0][l] + (gin ? gin[l] : 0.f);
    if (l < 3) gJ[fs][0][l] += gtw[fs][0][l];
    __syncthreads();
    if (live && l < 10) {
        const float* bi = d_betas_in ? d_betas_in + (int64_t)f * betas_in_stride : nullptr;
        float s = gpf[207 + l] + (bi ? bi[l] : 0.f);
        for (int j = 0; j < NJ; ++j)
            for (int c = 0; c < 3; ++c) s = fmaf(sp.J_shapedirs[(j * 3 + c) * 10 + l], gJ[fs][j][c], s);
        d_betas[(int64_t)f * 10 + l] = s;
    }
}

extern "C" int maed_smpl_chain_bwd(const maed_smpl_params* sp, const float* betas, const float* rotmat, const float* dA,
                                   const float* d_joints24, const float* dpf_dbeta, const float* d_rotmat_in, const float* d_betas_in,
                                   int64_t betas_in_stride, float* d_rotmat, float* d_betas, int F, void* stream) {
    MAED_CHECK_ARG(sp && betas && rotmat && dA && d_joints24 && dpf_dbeta && d_rotmat && d_betas, MAED_ERR_ARG, "smpl_chain_bwd: null pointer");
    MAED_CHECK_ARG(sp->J_template && sp->J_shapedirs && sp->parents, MAED_ERR_ARG, "smpl_chain_bwd: null SMPL parameter");
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(smpl_chain_bwd_par_kernel, dim3((F + SC_FPB - 1) / SC_FPB), dim3(64), 0, (hipStream_t)stream, *sp, betas, rotmat, dA,
                       d_joints24, dpf_dbeta, d_rotmat_in, d_betas_in, betas_in_stride, d_rotmat, d_betas, F);
    MAED_CHECK_LAUNCH("smpl_chain_bwd");
    return MAED_OK;
}

// ---- K11 backward -------------------------------------------------------------------------------------------------------------
// The forward (smpl.hip rot6d_pose_kernel; geometry.py:320-334,143-223,90-140) is re-evaluated on dual numbers carrying the six
// partial derivatives (forward-mode AD: the 6 -> 12 Jacobian costs 7x the forward, ~2 kFLOP per joint), then contracted
// with the incoming gradient.  Branches are taken on the VALUES, exactly as the forward takes them.
template <typename S>
__device__ __forceinline__ void rot6d_pose_eval(const S (&x)[6], S (&R)[9], S (&aa)[3]) {
    const S a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const S n1 = clamp_min(dsqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-6f);
    const S b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const S dot = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const S u[3] = {a2[0] - dot * b1[0], a2[1] - dot * b1[1], a2[2] - dot * b1[2]};
    const S n2 = clamp_min(dsqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-6f);
    const S b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const S b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = b1[r]; R[r * 3 + 1] = b2[r]; R[r * 3 + 2] = b3[r]; }
#define RT(a, b) R[(b) * 3 + (a)]
    const bool d2 = val(RT(2, 2)) < 1e-6f, d0d1 = val(RT(0, 0)) > val(RT(1, 1)), d0nd1 = val(RT(0, 0)) < -val(RT(1, 1));
    S q[4], t;
    if (d2 && d0d1) {
        t = 1.0f + RT(0, 0) - RT(1, 1) - RT(2, 2);
        q[0] = RT(1, 2) - RT(2, 1); q[1] = t; q[2] = RT(0, 1) + RT(1, 0); q[3] = RT(2, 0) + RT(0, 2);
    } else if (d2 && !d0d1) {
        t = 1.0f - RT(0, 0) + RT(1, 1) - RT(2, 2);
        q[0] = RT(2, 0) - RT(0, 2); q[1] = RT(0, 1) + RT(1, 0); q[2] = t; q[3] = RT(1, 2) + RT(2, 1);
    } else if (!d2 && d0nd1) {
        t = 1.0f - RT(0, 0) - RT(1, 1) + RT(2, 2);
        q[0] = RT(0, 1) - RT(1, 0); q[1] = RT(2, 0) + RT(0, 2); q[2] = RT(1, 2) + RT(2, 1); q[3] = t;
    } else {
        t = 1.0f + RT(0, 0) + RT(1, 1) + RT(2, 2);
        q[0] = t; q[1] = RT(1, 2) - RT(2, 1); q[2] = RT(2, 0) - RT(0, 2); q[3] = RT(0, 1) - RT(1, 0);
    }
#undef RT
    const S st = dsqrt(t);
    for (int k = 0; k < 4; ++k) q[k] = (q[k] / st) * 0.5f;
    const S ss = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const S sn = dsqrt(ss), cs = q[0];
    const S two_theta = (val(cs) < 0.0f ? datan2(-sn, -cs) : datan2(sn, cs)) * 2.0f;
    const S k = val(ss) > 0.0f ? two_theta / sn : constant<S>(2.0f);
    for (int c = 0; c < 3; ++c) aa[c] = zero_if_nan(q[1 + c] * k);    // geometry.py:86
}

__global__ void rot6d_pose_bwd_kernel(const float* __restrict__ x6, const float* __restrict__ d_rot, const float* __restrict__ d_aa,
                                      int64_t aa_stride, float* __restrict__ d_x6, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Dual<6> D;
    D x[6], R[9], aa[3];
    for (int k = 0; k < 6; ++k) x[k] = seed<6>(x6[i * 6 + k], k);
    rot6d_pose_eval<D>(x, R, aa);
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 9; ++k) {
        const float gr = d_rot[i * 9 + k];
        for (int m = 0; m < 6; ++m) g[m] = fmaf(gr, R[k].d[m], g[m]);
    }
    if (d_aa) {
        const float* ga = d_aa + (i / NJ) * aa_stride + (i % NJ) * 3;
        for (int c = 0; c < 3; ++c)
            for (int m = 0; m < 6; ++m) g[m] = fmaf(ga[c], aa[c].d[m], g[m]);
    }
    for (int m = 0; m < 6; ++m) d_x6[i * 6 + m] = g[m];
}

extern "C" int maed_rot6d_pose_bwd(const float* pose6d, const float* d_rotmat, const float* d_aa, int64_t aa_stride, float* d_pose6d,
                                   int64_t n_joints, void* stream) {
    MAED_CHECK_ARG(pose6d && d_rotmat && d_pose6d, MAED_ERR_ARG, "rot6d_pose_bwd: null pointer");
    if (n_joints <= 0) return MAED_OK;
    hipLaunchKernelGGL(rot6d_pose_bwd_kernel, dim3((unsigned)((n_joints + 127) / 128)), dim3(128), 0, (hipStream_t)stream, pose6d, d_rotmat, d_aa,
                       aa_stride, d_pose6d, n_joints);
    MAED_CHECK_LAUNCH("rot6d_pose_bwd");
    return MAED_OK;
}

// ---- K10 backward ---------------------------------------------------------------------------------------------------------------
// pose_j = base_j + sum_slot W_j[:, slot] pose_anc(j,slot):  walking j = 23..1, g_anc += W_j[:, slot]^T g_j (ancestors have
// smaller indices, so g_j is final when j is visited).  d_base = g.
// 16 lanes per frame, 4 frames per workgroup, g in LDS.  The 6*n_anc(j) gradient elements joint j feeds (distinct (ancestor, i) pairs) are
// updated by different lanes, each with one fmaf chain over o.
#define KB_FPB 4
__global__ __launch_bounds__(64) void ktd_chain_bwd_par_kernel(const float* __restrict__ w_anc, const float* __restrict__ d_pose,
                                                               const float* __restrict__ d_shape, const float* __restrict__ d_cam,
                                                               float* __restrict__ d_out, int64_t ld, int F) {
    __shared__ float g[KB_FPB][NJ * 6];
    const int l = threadIdx.x & 15, fs = threadIdx.x >> 4;
    const int f = blockIdx.x * KB_FPB + fs, fc = f < F ? f : F - 1;
    for (int i = l; i < NJ * 6; i += 16) g[fs][i] = d_pose[(int64_t)fc * NJ * 6 + i];
    __syncthreads();
    for (int j = NJ - 1; j >= 1; --j) {
        const int na = c_anc_cnt[j];
        const float* W = w_anc + 36 * c_anc_start[j];
        for (int t = l; t < 6 * na; t += 16) {
            const int sl = t / 6, i = t % 6;
            const int a = c_anc[c_anc_start[j] + sl];
            float s = g[fs][a * 6 + i];
            for (int o = 0; o < 6; ++o) s = fmaf(W[o * 6 * na + sl * 6 + i], g[fs][j * 6 + o], s);
            g[fs][a * 6 + i] = s;
        }
        __syncthreads();
    }
    if (f >= F) return;
    float* o = d_out + (int64_t)f * ld;
    for (int i = l; i < NJ * 6; i += 16) o[i] = g[fs][i];
    if (l < 10) o[144 + l] = d_shape ? d_shape[(int64_t)f * 10 + l] : 0.f;
    if (l < 3) o[154 + l] = d_cam ? d_cam[(int64_t)f * 3 + l] : 0.f;
}

// EIGHT lanes per ancestor-weight element: dW_j[o][6*slot+i] = sum_f d_base[f][6j+o] pose[f][6*anc+i], every lane over every eighth frame, folded with three
// shuffles; the last 157 elements are the bias gradient (column sums of d_out).  (One thread per element walked all F frames in one dependent chain of strided
// loads: 55 us for 128 frames, the longest kernel of the decoder's backward.)
#define KW_LANES 8
__global__ __launch_bounds__(128) void ktd_wanc_bwd_kernel(const float* __restrict__ pose, const float* __restrict__ d_out, int64_t ld, float* __restrict__ d_w_anc,
                                                           float* __restrict__ d_b_feat, int F) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / KW_LANES, part = t % KW_LANES;
    const bool live = e < MAED_KTD_W_ANC + KTD_OUT;          // (the lanes of a dead element still take part in the shuffles)
    float s = 0.f;
    if (live && e >= MAED_KTD_W_ANC) {
        const int c = e - MAED_KTD_W_ANC;
        for (int f = part; f < F; f += KW_LANES) s += d_out[(int64_t)f * ld + c];
    } else if (live) {
        int j = 1;
        while (36 * c_anc_start[j + 1] <= e) ++j;
        const int na = c_anc_cnt[j], tt = e - 36 * c_anc_start[j];
        const int o = tt / (6 * na), col = tt % (6 * na);
        const int a = c_anc[c_anc_start[j] + col / 6], i = col % 6;
        for (int f = part; f < F; f += KW_LANES) s = fmaf(d_out[(int64_t)f * ld + j * 6 + o], pose[(int64_t)f * NJ * 6 + a * 6 + i], s);
    }
#pragma unroll
    for (int m = 1; m < KW_LANES; m <<= 1) s += __shfl_xor(s, m);
    if (!live || part != 0) return;
    if (e >= MAED_KTD_W_ANC) d_b_feat[e - MAED_KTD_W_ANC] = s; else d_w_anc[e] = s;
}

extern "C" int maed_ktd_chain_bwd(const float* pose, const float* w_anc, const float* d_pose, const float* d_shape, const float* d_cam,
                                  float* d_out, int64_t ld_out, float* d_w_anc, float* d_b_feat, int F, void* stream) {
    MAED_CHECK_ARG(pose && w_anc && d_pose && d_out && d_w_anc && d_b_feat, MAED_ERR_ARG, "ktd_chain_bwd: null pointer");
    MAED_CHECK_ARG(ld_out >= KTD_OUT, MAED_ERR_SHAPE, "ktd_chain_bwd: ld_out=%lld < 157", (long long)ld_out);
    hipStream_t s = (hipStream_t)stream;
    if (F > 0)
        hipLaunchKernelGGL(ktd_chain_bwd_par_kernel, dim3((F + KB_FPB - 1) / KB_FPB), dim3(64), 0, s, w_anc, d_pose, d_shape, d_cam, d_out, ld_out, F);
    hipLaunchKernelGGL(ktd_wanc_bwd_kernel, dim3(((MAED_KTD_W_ANC + KTD_OUT) * KW_LANES + 127) / 128), dim3(128), 0, s, pose, d_out, ld_out, d_w_anc, d_b_feat, F);
    MAED_CHECK_LAUNCH("ktd_chain_bwd");
    return MAED_OK;
}

// ---- pack / unpack of the 26 regressor weights ------------------------------------------------------------------------------------
// packed row r < 144: regressor j = r/6, output o = r%6 (row length hidden + 6*n_anc(j)); 144..153 decshape; 154..156 deccam
__device__ __forceinline__ void ktd_row(int r, int hidden, int& j, int& o, int& ld) {
    if (r < 144) { j = r / 6; o = r % 6; ld = hidden + 6 * c_anc_cnt[j]; }
    else if (r < 154) { j = 24; o = r - 144; ld = hidden; }
    else { j = 25; o = r - 154; ld = hidden; }
}

template <bool UNPACK>
__global__ __launch_bounds__(256) void ktd_pack_kernel(maed_ktd_ptrs t, int hidden, float* __restrict__ w_feat, float* __restrict__ b_feat,
                                                       float* __restrict__ w_anc) {
    const int r = blockIdx.x;
    if (r < KTD_OUT) {
        int j, o, ld;
        ktd_row(r, hidden, j, o, ld);
        for (int c = threadIdx.x; c < hidden; c += 256) {
            if (UNPACK) t.gw[j][(int64_t)o * ld + c] += w_feat[(int64_t)r * hidden + c];
            else w_feat[(int64_t)r * hidden + c] = t.w[j][(int64_t)o * ld + c];
        }
        if (threadIdx.x == 0) {
            if (UNPACK) t.gb[j][o] += b_feat[r];
            else b_feat[r] = t.b[j][o];
        }
        if (r < 144 && threadIdx.x < 6 * c_anc_cnt[j]) {   // the ancestor columns of this row
            const int na = c_anc_cnt[j];
            const int e = 36 * c_anc_start[j] + o * 6 * na + threadIdx.x;
            if (UNPACK) t.gw[j][(int64_t)o * ld + hidden + threadIdx.x] += w_anc[e];
            else w_anc[e] = t.w[j][(int64_t)o * ld + hidden + threadIdx.x];
        }
    }
}

static int ktd_ptrs_ok(const maed_ktd_ptrs* t, bool grads) {
    for (int j = 0; j < 26; ++j)
        if (grads ? !(t->gw[j] && t->gb[j]) : !(t->w[j] && t->b[j])) return 0;
    return 1;
}

extern "C" int maed_ktd_pack(const maed_ktd_ptrs* t, int hidden, float* w_feat, float* b_feat, float* w_anc, void* stream) {
    MAED_CHECK_ARG(t && w_feat && b_feat && w_anc && ktd_ptrs_ok(t, false), MAED_ERR_ARG, "ktd_pack: null pointer");
    MAED_CHECK_ARG(hidden > 0, MAED_ERR_SHAPE, "ktd_pack: hidden=%d", hidden);
    hipLaunchKernelGGL((ktd_pack_kernel<false>), dim3(KTD_OUT), dim3(256), 0, (hipStream_t)stream, *t, hidden, w_feat, b_feat, w_anc);
    MAED_CHECK_LAUNCH("ktd_pack");
    return MAED_OK;
}

extern "C" int maed_ktd_unpack_add(const maed_ktd_ptrs* t, int hidden, const float* d_w_feat, const float* d_b_feat, const float* d_w_anc,
                                   void* stream) {
    MAED_CHECK_ARG(t && d_w_feat && d_b_feat && d_w_anc && ktd_ptrs_ok(t, true), MAED_ERR_ARG, "ktd_unpack_add: null pointer");
    MAED_CHECK_ARG(hidden > 0, MAED_ERR_SHAPE, "ktd_unpack_add: hidden=%d", hidden);
    hipLaunchKernelGGL((ktd_pack_kernel<true>), dim3(KTD_OUT), dim3(256), 0, (hipStream_t)stream, *t, hidden, (float*)d_w_feat, (float*)d_b_feat,
                       (float*)d_w_anc);
    MAED_CHECK_LAUNCH("ktd_unpack_add");
    return MAED_OK;
}
