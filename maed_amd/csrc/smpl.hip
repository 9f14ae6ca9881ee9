// Kinematic Topology Decoder tail: joint chain (K10), 6D->rotmat->axis-angle (K11), SMPL linear blend
// skinning (K12), joint regressor GEMM on f32 MFMA (K13), integer joint gather (K14, bit-exact) and
// weak-perspective projection (K15).  Everything here is fp32: these are SMPL parameters (1e-3 bar).
#include "common.cuh"
#include "ktd_tables.cuh"

// ---- K10: pose[f][6j+o] = base[f][6j+o] + sum_{slot,i} W_j[o][6*slot+i] * pose[f][6*anc(j,slot)+i] ------------
// The 6 outputs of a joint on 6 lanes (8 lanes per frame, 8 frames per 64-thread workgroup), the pose in LDS: every output is one fmaf chain over its
// ancestors' elements in slot order, the dependent chain per frame is 570 FMAs (a thread-per-frame kernel: 3420, with the pose in scratch; it
// was 4x slower on MI355X -- profiles/r02_call2_steady_*.csv -- and is gone).
#define KC_FPB 8
__global__ __launch_bounds__(64) void ktd_chain_par_kernel(const float* __restrict__ base, const float* __restrict__ w_anc, float* __restrict__ pose, int F) {
    __shared__ float ps[KC_FPB][NJ * 6];
    const int o = threadIdx.x & 7, fs = threadIdx.x >> 3;
    const int f = blockIdx.x * KC_FPB + fs, fc = f < F ? f : F - 1;
    for (int i = o; i < NJ * 6; i += 8) ps[fs][i] = base[(int64_t)fc * NJ * 6 + i];
    __syncthreads();
    for (int j = 1; j < NJ; ++j) {
        if (o < 6) {
            const int na = c_anc_cnt[j];
            const float* W = w_anc + 36 * c_anc_start[j] + o * 6 * na;
            float s = ps[fs][j * 6 + o];
            for (int sl = 0; sl < na; ++sl) {
                const int a = c_anc[c_anc_start[j] + sl];
                for (int i = 0; i < 6; ++i) s = fmaf(W[sl * 6 + i], ps[fs][a * 6 + i], s);
            }
            ps[fs][j * 6 + o] = s;          // nobody else reads element (j, o) in this step: ancestors have smaller indices
        }
        __syncthreads();
    }
    if (f < F)
        for (int i = o; i < NJ * 6; i += 8) pose[(int64_t)f * NJ * 6 + i] = ps[fs][i];
}


extern "C" int maed_ktd_chain_fwd(const float* base, const float* w_anc, float* pose, int F, void* stream) {
    MAED_CHECK_ARG(base && w_anc && pose, MAED_ERR_ARG, "ktd_chain_fwd: null pointer");
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(ktd_chain_par_kernel, dim3((F + KC_FPB - 1) / KC_FPB), dim3(64), 0, (hipStream_t)stream, base, w_anc, pose, F);
    MAED_CHECK_LAUNCH("ktd_chain_fwd");
    return MAED_OK;
}

// ---- K11 --------------------------------------------------------------------------------------------------
__global__ void rot6d_pose_kernel(const float* __restrict__ x6, float* __restrict__ rotmat, float* __restrict__ aa, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* x = x6 + i * 6;
    // geometry.py:320-334: x.view(-1,3,2): a1 = x[:, :, 0] = (x0,x2,x4), a2 = x[:, :, 1] = (x1,x3,x5)
    float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    float n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    n1 = fmaxf(n1, 1e-6f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float dot = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float u[3] = {a2[0] - dot * b1[0], a2[1] - dot * b1[1], a2[2] - dot * b1[2]};
    float n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    n2 = fmaxf(n2, 1e-6f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    float R[3][3];
    for (int r = 0; r < 3; ++r) { R[r][0] = b1[r]; R[r][1] = b2[r]; R[r][2] = b3[r]; }
    if (rotmat) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rotmat[i * 9 + r * 3 + c] = R[r][c];
    if (!aa) return;
    // geometry.py:143-223 with rmat_t = R^T: t(a,b) = R[b][a]
#define RT(a, b) R[b][a]
    const bool d2 = RT(2, 2) < 1e-6f, d0d1 = RT(0, 0) > RT(1, 1), d0nd1 = RT(0, 0) < -RT(1, 1);
    float q[4], t;
    if (d2 && d0d1) {
        t = 1 + RT(0, 0) - RT(1, 1) - RT(2, 2);
        q[0] = RT(1, 2) - RT(2, 1); q[1] = t; q[2] = RT(0, 1) + RT(1, 0); q[3] = RT(2, 0) + RT(0, 2);
    } else if (d2 && !d0d1) {
        t = 1 - RT(0, 0) + RT(1, 1) - RT(2, 2);
        q[0] = RT(2, 0) - RT(0, 2); q[1] = RT(0, 1) + RT(1, 0); q[2] = t; q[3] = RT(1, 2) + RT(2, 1);
    } else if (!d2 && d0nd1) {
        t = 1 - RT(0, 0) - RT(1, 1) + RT(2, 2);
        q[0] = RT(0, 1) - RT(1, 0); q[1] = RT(2, 0) + RT(0, 2); q[2] = RT(1, 2) + RT(2, 1); q[3] = t;
    } else {
        t = 1 + RT(0, 0) + RT(1, 1) + RT(2, 2);
        q[0] = t; q[1] = RT(1, 2) - RT(2, 1); q[2] = RT(2, 0) - RT(0, 2); q[3] = RT(0, 1) - RT(1, 0);
    }
#undef RT
    const float st = sqrtf(t);
    for (int k = 0; k < 4; ++k) q[k] = (q[k] / st) * 0.5f;
    // geometry.py:90-140
    const float ss = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const float sn = sqrtf(ss), cs = q[0];
    const float two_theta = 2.0f * (cs < 0.0f ? atan2f(-sn, -cs) : atan2f(sn, cs));
    const float k = ss > 0.0f ? two_theta / sn : 2.0f;
    for (int c = 0; c < 3; ++c) {
        float v = q[1 + c] * k;
        if (v != v) v = 0.0f;  // geometry.py:86 aa[isnan(aa)] = 0
        aa[i * 3 + c] = v;
    }
}

extern "C" int maed_rot6d_pose_fwd(const float* pose6d, float* rotmat, float* angle_axis, int64_t n_joints, void* stream) {
    MAED_CHECK_ARG(pose6d && (rotmat || angle_axis), MAED_ERR_ARG, "rot6d_pose_fwd: null pointer");
    if (n_joints <= 0) return MAED_OK;
    hipLaunchKernelGGL(rot6d_pose_kernel, dim3((unsigned)((n_joints + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pose6d, rotmat, angle_axis, n_joints);
    MAED_CHECK_LAUNCH("rot6d_pose_fwd");
    return MAED_OK;
}

// ---- K12: SMPL LBS (smplx.lbs.lbs, pose2rot=False) ---------------------------------------
// kernel A: the kinematic chain -> A (24 x 3x4 skinning transforms) + posed joints.  16 lanes per frame, 4 frames per workgroup, J / Rw / tw in LDS: the
// 72 rest-pose joint coordinates, the 12 outputs of each joint's transform and the 24 x 15 outputs are spread over the lanes, the tree is walked
// joint by joint (any parent table).
#define LC_FPB 4
__global__ __launch_bounds__(64) void lbs_chain_par_kernel(maed_smpl_params sp, const float* __restrict__ betas, const float* __restrict__ rotmat,
                                                           float* __restrict__ joints24, float* __restrict__ A, int F) {
    __shared__ float sJ[LC_FPB][NJ][3], sRw[LC_FPB][NJ][9], stw[LC_FPB][NJ][3];
    const int l = threadIdx.x & 15, fs = threadIdx.x >> 4;
    const int f = blockIdx.x * LC_FPB + fs, fc = f < F ? f : F - 1;
    const float* b = betas + (int64_t)fc * 10;
    for (int t = l; t < NJ * 3; t += 16) {
        float s = sp.J_template[t];
        for (int k = 0; k < 10; ++k) s = fmaf(sp.J_shapedirs[t * 10 + k], b[k], s);
        sJ[fs][t / 3][t % 3] = s;
    }
    __syncthreads();
    const float* R = rotmat + (int64_t)fc * NJ * 9;
    for (int j = 0; j < NJ; ++j) {
        const int p = sp.parents[j];
        const float* Rj = R + j * 9;
        if (l < 12) {
            if (p < 0) {
                if (l < 9) sRw[fs][j][l] = Rj[l];
                else stw[fs][j][l - 9] = sJ[fs][j][l - 9];
            } else if (l < 9) {
                const int r = l / 3, c = l % 3;
                sRw[fs][j][l] = sRw[fs][p][r * 3 + 0] * Rj[0 * 3 + c] + sRw[fs][p][r * 3 + 1] * Rj[1 * 3 + c] + sRw[fs][p][r * 3 + 2] * Rj[2 * 3 + c];
            } else {
                const int r = l - 9;
                const float rel[3] = {sJ[fs][j][0] - sJ[fs][p][0], sJ[fs][j][1] - sJ[fs][p][1], sJ[fs][j][2] - sJ[fs][p][2]};
                stw[fs][j][r] = sRw[fs][p][r * 3 + 0] * rel[0] + sRw[fs][p][r * 3 + 1] * rel[1] + sRw[fs][p][r * 3 + 2] * rel[2] + stw[fs][p][r];
            }
        }
        __syncthreads();
    }
    if (f >= F) return;
    for (int t = l; t < NJ * 3; t += 16) {
        const int j = t / 3, r = t % 3;
        float* Aj = A + ((int64_t)f * NJ + j) * 12;
        for (int c = 0; c < 3; ++c) Aj[r * 4 + c] = sRw[fs][j][r * 3 + c];
        Aj[r * 4 + 3] = stw[fs][j][r] - (sRw[fs][j][r * 3 + 0] * sJ[fs][j][0] + sRw[fs][j][r * 3 + 1] * sJ[fs][j][1] + sRw[fs][j][r * 3 + 2] * sJ[fs][j][2]);
        joints24[((int64_t)f * NJ + j) * 3 + r] = stw[fs][j][r];
    }
}

// kernel B: thread per vertex, LBS_FB frames per workgroup so posedirs (17 MB) is streamed once per LBS_FB frames
// (measured at 128 frames, scripts/lbs_micro.py: 1 / 2 / 4 / 8 frames per workgroup = 319 / 279 / 164 / 194 us with the pose feature as [frame][k])
template <int LBS_FB>
__global__ __launch_bounds__(256) void lbs_skin_kernel(maed_smpl_params sp, const float* __restrict__ betas, const float* __restrict__ rotmat,
                                                       const float* __restrict__ A, float* __restrict__ verts,
                                                       float* __restrict__ v_posed, int F) {
    // pose feature as [k][frame]: the LBS_FB weights of one k are ONE 16- / 32-byte LDS read (broadcast) instead of LBS_FB four-byte ones
    __shared__ __attribute__((aligned(16))) float s_pf[208][LBS_FB];
    __shared__ float s_A[LBS_FB][NJ * 12];
    __shared__ float s_b[LBS_FB][10];
    const int f0 = blockIdx.y * LBS_FB;
    for (int i = threadIdx.x; i < LBS_FB * 207; i += 256) {
        const int fb = i / 207, k = i % 207;
        const int f = min(f0 + fb, F - 1);
        const int j = 1 + k / 9, rc = k % 9;
        s_pf[k][fb] = rotmat[((int64_t)f * NJ + j) * 9 + rc] - ((rc == 0 || rc == 4 || rc == 8) ? 1.f : 0.f);
    }
    for (int i = threadIdx.x; i < LBS_FB * NJ * 12; i += 256) {
        const int fb = i / (NJ * 12), k = i % (NJ * 12);
        s_A[fb][k] = A[(int64_t)min(f0 + fb, F - 1) * NJ * 12 + k];
    }
    if (threadIdx.x < LBS_FB * 10) s_b[threadIdx.x / 10][threadIdx.x % 10] = betas[(int64_t)min(f0 + threadIdx.x / 10, F - 1) * 10 + threadIdx.x % 10];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= NV) return;
    float vp[LBS_FB][3];
    for (int c = 0; c < 3; ++c) {
        const float vt = sp.v_template[v * 3 + c];
        float sd[10];
        for (int l = 0; l < 10; ++l) sd[l] = sp.shapedirs[(v * 3 + c) * 10 + l];
#pragma unroll
        for (int fb = 0; fb < LBS_FB; ++fb) {
            float s = vt;
            for (int l = 0; l < 10; ++l) s = fmaf(sd[l], s_b[fb][l], s);
            vp[fb][c] = s;
        }
    }
    float po[LBS_FB][3];
#pragma unroll
    for (int fb = 0; fb < LBS_FB; ++fb) { po[fb][0] = 0.f; po[fb][1] = 0.f; po[fb][2] = 0.f; }
    // (`#pragma unroll 9` here -- nine rows of posedirs in flight per lane -- measured 285 us at 4 frames per workgroup against 157 rolled)
    for (int k = 0; k < 207; ++k) {
        const float* pd = sp.posedirs + (int64_t)k * (NV * 3) + v * 3;
        const float p0 = pd[0], p1 = pd[1], p2 = pd[2];
        float wk[LBS_FB];
#pragma unroll
        for (int q = 0; q < LBS_FB / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4*>(&s_pf[k][4 * q]);
            wk[4 * q] = w4.x; wk[4 * q + 1] = w4.y; wk[4 * q + 2] = w4.z; wk[4 * q + 3] = w4.w;
        }
#pragma unroll
        for (int fb = 0; fb < LBS_FB; ++fb) {
            const float w = wk[fb];
            po[fb][0] = fmaf(w, p0, po[fb][0]); po[fb][1] = fmaf(w, p1, po[fb][1]); po[fb][2] = fmaf(w, p2, po[fb][2]);
        }
    }
    float w[NJ];
    for (int j = 0; j < NJ; ++j) w[j] = sp.lbs_weights[v * NJ + j];
#pragma unroll
    for (int fb = 0; fb < LBS_FB; ++fb) {
        if (f0 + fb >= F) break;
        const float x = vp[fb][0] + po[fb][0], y = vp[fb][1] + po[fb][1], z = vp[fb][2] + po[fb][2];
        if (v_posed) {   // kept for the backward pass (maed_smpl_skin_bwd) instead of a second sweep over posedirs
            float* q = v_posed + ((int64_t)(f0 + fb) * NV + v) * 3;
            q[0] = x; q[1] = y; q[2] = z;
        }
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j)
            for (int e = 0; e < 12; ++e) T[e] = fmaf(w[j], s_A[fb][j * 12 + e], T[e]);
        float* o = verts + ((int64_t)(f0 + fb) * NV + v) * 3;
        o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
}

// ---- K12 on the matrix cores (round 5) --------------------------------------------------------------------------------------------
// v_posed = v_template + shapedirs . beta + posedirs^T . (R[1:] - I) is ONE product per frame group: [frames x 218] . [218 x 20670] with the feature row
// (207 pose features | 10 betas | 1) -- on v_mfma_f32_32x32x2_f32 (exact fp32: an fmaf chain in k order; 157 TF/s) instead of 207 rolled VALU iterations per vertex
// thread with one wave per SIMD (lbs_skin_kernel: 140-150 us at 128 frames).  A wave owns one 32-column tile of the 20670 columns for up to 128 frames (four
// accumulators); its B operand -- posedirs rows, the transposed shape directions, the template -- is streamed ONCE per frame group (17 MB for the whole launch) as
// 128-byte row pieces, its A operand is read straight from the rotation matrices (k-th pose feature of frame f = rotmat[f][9 + k] minus the identity's diagonal).
// The skinning itself (24 x 12 blend per vertex and frame) follows as a streaming kernel over v_posed.
#define LB_KF 224                          // feature rows: 207 pose features, 10 betas, 1 (template), 6 zero rows (whole groups of 8 k-pairs)
#define LB_FG 64                           // frames per workgroup (two 32-frame MFMA tiles per wave): 58 KB of LDS, two workgroups per CU
#define LB_LD (LB_FG + 1)                  // LDS row stride of the [k][frame] feature image (floats): staging writes walk k, fragment reads walk the frame
__global__ __launch_bounds__(256) void lbs_blend_mfma_kernel(maed_smpl_params sp, const float* __restrict__ betas, const float* __restrict__ rotmat,
                                                            float* __restrict__ v_posed, int F) {
    MAED_DYN_SHARED(float, s_feat);        // [LB_KF][LB_LD]: feature k of frame f0 + f.  (First version: A operands straight from rotmat -- 32 cache lines per load
                                           // instruction, 416 of them per wave: 158 us against the VALU kernel's 137; second: one dependent B load per k-pair,
                                           // nothing in flight behind it: 122 us.  Now eight B loads are issued ahead of the 16 MFMAs that consume them.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int f0 = blockIdx.y * LB_FG, nf = min(LB_FG, F - f0);
    // staging: thread -> (frame tid >> 2, k = (tid & 3) + 4 q): no integer division in the loop; frames past F stage a duplicate of the last frame (their results
    // are never stored), so every load is unconditional -- a predicate made hipcc branch around each load and wait for it (52 serial round trips per thread)
    {
        const int f = tid >> 2, kq = tid & 3;
        const float* rrow = rotmat + (int64_t)min(f0 + f, F - 1) * (NJ * 9) + 9;
        float v[52];
#pragma unroll
        for (int q = 0; q < 52; ++q) { const int k = kq + 4 * q; v[q] = rrow[k < 207 ? k : 206]; }
#pragma unroll
        for (int q = 0; q < 52; ++q) {
            const int k = kq + 4 * q, m9 = k % 9;
            if (k < 207) s_feat[k * LB_LD + f] = v[q] - ((m9 == 0 || m9 == 4 || m9 == 8) ? 1.f : 0.f);
        }
    }
    for (int i = tid; i < LB_FG * (LB_KF - 207); i += 256) {
        const int f = i / (LB_KF - 207), l = i - f * (LB_KF - 207);
        s_feat[(207 + l) * LB_LD + f] = l < 10 ? betas[(int64_t)min(f0 + f, F - 1) * 10 + l] : l == 10 ? 1.f : 0.f;
    }
    __syncthreads();
    const int c0 = (blockIdx.x * 4 + wave) * 32;
    if (c0 >= NV * 3) return;
    const int col = min(c0 + l31, NV * 3 - 1);
    f32x16_t acc[LB_FG / 32];
#pragma unroll
    for (int t = 0; t < LB_FG / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // B operand: groups of eight k-pairs, the NEXT group's loads issued before this group's 16 MFMAs (two register sets, named: no runtime-indexed arrays);
    // every frame tile is computed (a partial frame group multiplies duplicates): no branch between the MFMAs
#define LB_LOAD(dst_, kk0_) _Pragma("unroll") for (int u = 0; u < 8; ++u) { const int k = 2 * ((kk0_) + u) + hi; \
        const float* src__ = k < 207 ? sp.posedirs + (int64_t)k * (NV * 3) + col : k < 217 ? sp.shapedirs + (int64_t)col * 10 + (k - 207) : sp.v_template + col; \
        dst_[u] = *src__; }     /* (k > 217: feature 0) */
#define LB_MMA(src_, kk0_) _Pragma("unroll") for (int u = 0; u < 8; ++u) { const float* fa = s_feat + (2 * ((kk0_) + u) + hi) * LB_LD + l31; \
        _Pragma("unroll") for (int t = 0; t < LB_FG / 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[32 * t], src_[u], acc[t], 0, 0, 0); }
    float b0[8], b1[8];
    LB_LOAD(b0, 0)
    for (int kk0 = 0; kk0 < LB_KF / 2; kk0 += 16) {          // LB_KF / 2 = 112 = 7 x 16
        LB_LOAD(b1, kk0 + 8)
        LB_MMA(b0, kk0)
        if (kk0 + 16 < LB_KF / 2) LB_LOAD(b0, kk0 + 16)
        LB_MMA(b1, kk0 + 8)
    }
#undef LB_LOAD
#undef LB_MMA
    if (c0 + l31 >= NV * 3) return;
#pragma unroll
    for (int t = 0; t < LB_FG / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (f < nf) v_posed[(int64_t)(f0 + f) * (NV * 3) + c0 + l31] = acc[t][r];
        }
}

// verts[f][v] = (sum_j w[v][j] A[f][j]) . [v_posed[f][v]; 1]: thread per vertex, FB frames per workgroup (the weights of a vertex are read once per FB frames).
// v_posed may alias verts (inference: the blend is written into the output buffer and skinned in place -- a thread reads its own three values before it writes them).
template <int FB>
__global__ __launch_bounds__(256) void lbs_skin_posed_kernel(maed_smpl_params sp, const float* __restrict__ A, const float* v_posed, float* verts, int F) {
    __shared__ float s_A[FB][NJ * 12];
    const int f0 = blockIdx.y * FB;
    for (int i = threadIdx.x; i < FB * NJ * 12; i += 256) {
        const int fb = i / (NJ * 12), k = i % (NJ * 12);
        s_A[fb][k] = A[(int64_t)min(f0 + fb, F - 1) * NJ * 12 + k];
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= NV) return;
    float w[NJ];
    for (int j = 0; j < NJ; ++j) w[j] = sp.lbs_weights[v * NJ + j];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
        if (f0 + fb >= F) break;
        const float* q = v_posed + ((int64_t)(f0 + fb) * NV + v) * 3;
        const float x = q[0], y = q[1], z = q[2];
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j)
            for (int e = 0; e < 12; ++e) T[e] = fmaf(w[j], s_A[fb][j * 12 + e], T[e]);
        float* o = verts + ((int64_t)(f0 + fb) * NV + v) * 3;
        o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
}

extern "C" int maed_smpl_lbs_fwd(const maed_smpl_params* sp, const float* betas, const float* rotmat, float* verts,
                                 float* joints24, float* scratch_A, float* v_posed, int F, void* stream) {
    MAED_CHECK_ARG(sp && betas && rotmat && verts && joints24 && scratch_A, MAED_ERR_ARG, "smpl_lbs_fwd: null pointer");
    MAED_CHECK_ARG(sp->v_template && sp->shapedirs && sp->posedirs && sp->J_template && sp->J_shapedirs && sp->lbs_weights && sp->parents,
                   MAED_ERR_ARG, "smpl_lbs_fwd: null SMPL parameter");
    if (F <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lbs_chain_par_kernel, dim3((F + LC_FPB - 1) / LC_FPB), dim3(64), 0, s, *sp, betas, rotmat, joints24, scratch_A, F);
    if (maed_opt(MAED_OPT_LBS_FRAMES) == 0) {      // default (round 5): the blend on the fp32 matrix cores, then the skinning pass (MAED_OPT_LBS_FRAMES = 4 / 8 / 16: the VALU kernel)
        float* vp = v_posed ? v_posed : verts;      // inference: blend into the output buffer, skinned in place
        constexpr size_t lds = (size_t)LB_KF * LB_LD * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)lbs_blend_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
        hipLaunchKernelGGL(lbs_blend_mfma_kernel, dim3((NV * 3 + 127) / 128, (F + LB_FG - 1) / LB_FG), dim3(256), lds, s, *sp, betas, rotmat, vp, F);
        hipLaunchKernelGGL(lbs_skin_posed_kernel<4>, dim3((NV + 255) / 256, (F + 3) / 4), dim3(256), 0, s, *sp, scratch_A, vp, verts, F);
        MAED_CHECK_LAUNCH("smpl_lbs_fwd");
        return MAED_OK;
    }
    const int fb = maed_opt(MAED_OPT_LBS_FRAMES);     // frames per workgroup (4 / 8 / 16 = 157 / 183 / 138 us per forward at 128 frames)
    if (fb == 8) hipLaunchKernelGGL(lbs_skin_kernel<8>, dim3((NV + 255) / 256, (F + 7) / 8), dim3(256), 0, s, *sp, betas, rotmat, scratch_A, verts, v_posed, F);
    else if (fb == 4) hipLaunchKernelGGL(lbs_skin_kernel<4>, dim3((NV + 255) / 256, (F + 3) / 4), dim3(256), 0, s, *sp, betas, rotmat, scratch_A, verts, v_posed, F);
    else hipLaunchKernelGGL(lbs_skin_kernel<16>, dim3((NV + 255) / 256, (F + 15) / 16), dim3(256), 0, s, *sp, betas, rotmat, scratch_A, verts, v_posed, F);
    MAED_CHECK_LAUNCH("smpl_lbs_fwd");
    return MAED_OK;
}

// ---- K13: joint regressor  out[f][j][c] = sum_v Jreg[j][v] verts[f][v][c]  on v_mfma_f32_32x32x2_f32 ---------------
// GEMM view: A = Jreg (J<=32 rows x 6890), B[k=v][n = 3*fl + c] for a group of 10 frames (30 of 32 columns),
// K split into 256-vertex chunks, one wave per (chunk, frame group); both operands staged through LDS with
// coalesced loads; exact-f32 fma chains (MI355X_MICROARCH: f32-input MFMA == fmaf chain, bitwise).
#define JR_KC 256
#define JR_FG 10
__global__ __launch_bounds__(64) void joint_regress_kernel(const float* __restrict__ Jreg, int J, const float* __restrict__ verts,
                                                           float* __restrict__ out, int F) {
    __shared__ float Js[32][JR_KC + 1];
    __shared__ float Bs[JR_FG][JR_KC * 3];
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int v0 = blockIdx.x * JR_KC, f0 = blockIdx.y * JR_FG;
    const int kc = min(JR_KC, NV - v0);
    for (int i = lane; i < 32 * JR_KC; i += 64) {
        const int j = i / JR_KC, k = i % JR_KC;
        Js[j][k] = (j < J && k < kc) ? Jreg[(int64_t)j * NV + v0 + k] : 0.f;
    }
    for (int fl = 0; fl < JR_FG; ++fl) {
        const int f = f0 + fl;
        for (int i = lane; i < JR_KC * 3; i += 64)
            Bs[fl][i] = (f < F && i < kc * 3) ? verts[((int64_t)f * NV + v0) * 3 + i] : 0.f;
    }
    __syncthreads();
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fl = l31 / 3, c = l31 % 3;  // column n = l31 -> (frame, coord); n = 30, 31 idle
    for (int s = 0; s < JR_KC / 2; ++s) {
        const int k = 2 * s + hi;
        const float a = Js[l31][k];
        const float b = (l31 < 30) ? Bs[fl][k * 3 + c] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // D: col n = l31, row j = (r&3) + 8*(r>>2) + 4*hi
    if (l31 < 30 && f0 + fl < F) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (j < J) atomicAdd(out + ((int64_t)(f0 + fl) * J + j) * 3 + c, acc[r]);
        }
    }
}

// The same regression through the regressor's NON-ZEROS: SMPL's joint regressors are sparse (J_regressor_extra: 9 rows, a few dozen vertices each -- the dense
// product above spends 67 us per step on 6890 columns of which ~270 are non-zero).  CSR built once per regressor by the host (SMPL.regressor_csr); thread per
// (frame, joint, coordinate), the row's non-zeros in ascending vertex order (the dense fma chain without its zero terms).
__global__ __launch_bounds__(256) void joint_regress_csr_kernel(const int* __restrict__ rowptr, const int* __restrict__ cols, const float* __restrict__ vals, int J,
                                                                const float* __restrict__ verts, float* __restrict__ out, int F) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F * J * 3) return;
    const int c = i % 3, j = (i / 3) % J, f = i / (3 * J);
    const float* v = verts + (int64_t)f * NV * 3 + c;
    float s = 0.f;
    for (int k = rowptr[j]; k < rowptr[j + 1]; ++k) s = fmaf(vals[k], v[cols[k] * 3], s);
    out[i] = s;
}

extern "C" int maed_joint_regress_csr_fwd(const int32_t* rowptr, const int32_t* cols, const float* vals, int J, const float* verts, float* out, int F, void* stream) {
    MAED_CHECK_ARG(rowptr && cols && vals && verts && out, MAED_ERR_ARG, "joint_regress_csr_fwd: null pointer");
    MAED_CHECK_ARG(J > 0 && J <= 64, MAED_ERR_SHAPE, "joint_regress_csr_fwd: J=%d must be in 1..64", J);
    if (F <= 0) return MAED_OK;
    hipLaunchKernelGGL(joint_regress_csr_kernel, dim3((F * J * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const int*)rowptr, (const int*)cols, vals, J, verts, out, F);
    MAED_CHECK_LAUNCH("joint_regress_csr_fwd");
    return MAED_OK;
}

extern "C" int maed_joint_regress_fwd(const float* Jreg, int J, const float* verts, float* out, int F, void* stream) {
    MAED_CHECK_ARG(Jreg && verts && out, MAED_ERR_ARG, "joint_regress_fwd: null pointer");
    MAED_CHECK_ARG(J > 0 && J <= 32, MAED_ERR_SHAPE, "joint_regress_fwd: J=%d must be in 1..32", J);
    if (F <= 0) return MAED_OK;
    hipStream_t s = (hipStream_t)stream;
    MAED_HIP(hipMemsetAsync(out, 0, (size_t)F * J * 3 * sizeof(float), s), "memset");
    hipLaunchKernelGGL(joint_regress_kernel, dim3((NV + JR_KC - 1) / JR_KC, (F + JR_FG - 1) / JR_FG), dim3(64), 0, s, Jreg, J, verts, out, F);
    MAED_CHECK_LAUNCH("joint_regress_fwd");
    return MAED_OK;
}

// ---- K14 + K15 -------------------------------------------------------------------------------------------------
__global__ void joints_project_kernel(const float* __restrict__ joints24, const float* __restrict__ verts, const int64_t* __restrict__ extra_ids,
                                      const float* __restrict__ extra9, const int64_t* __restrict__ joint_map, const float* __restrict__ cam,
                                      const float* __restrict__ jover, int Jo, float* __restrict__ kp3d, float* __restrict__ kp2d, int F) {
    const int Jn = jover ? Jo : 49;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)F * Jn) return;
    const int64_t f = i / Jn; const int jj = (int)(i % Jn);
    float p[3];
    if (jover) {
        for (int c = 0; c < 3; ++c) p[c] = jover[i * 3 + c];
    } else {
        const int64_t idx = joint_map[jj];  // smpl.py:98-99: cat(45 smplx joints, 9 extra)[joint_map]
        const float* src;
        if (idx < 24) src = joints24 + (f * 24 + idx) * 3;
        else if (idx < 45) src = verts + (f * NV + extra_ids[idx - 24]) * 3;
        else src = extra9 + (f * 9 + (idx - 45)) * 3;
        for (int c = 0; c < 3; ++c) p[c] = src[c];
    }
    for (int c = 0; c < 3; ++c) kp3d[i * 3 + c] = p[c];
    // spin.py:113-157
    const float* cm = cam + f * 3;
    const float tz = 2.0f * 5000.0f / (224.0f * cm[0] + 1e-9f);
    const float X = p[0] + cm[1], Y = p[1] + cm[2], Z = p[2] + tz;
    kp2d[i * 2 + 0] = (5000.0f * (X / Z)) / 112.0f;
    kp2d[i * 2 + 1] = (5000.0f * (Y / Z)) / 112.0f;
}

extern "C" int maed_smpl_joints_project_fwd(const float* joints24, const float* verts, const int64_t* extra_vertex_ids, const float* extra9,
                                            const int64_t* joint_map, const float* cam, const float* joints_override, int Jo, float* kp3d,
                                            float* kp2d, int F, void* stream) {
    MAED_CHECK_ARG(cam && kp3d && kp2d, MAED_ERR_ARG, "smpl_joints_project_fwd: null pointer");
    MAED_CHECK_ARG(joints_override || (joints24 && verts && extra_vertex_ids && extra9 && joint_map), MAED_ERR_ARG, "smpl_joints_project_fwd: null joint source");
    if (F <= 0) return MAED_OK;
    const int64_t n = (int64_t)F * (joints_override ? Jo : 49);
    hipLaunchKernelGGL(joints_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, joints24, verts, extra_vertex_ids, extra9,
                       joint_map, cam, joints_override, Jo, kp3d, kp2d, F);
    MAED_CHECK_LAUNCH("smpl_joints_project_fwd");
    return MAED_OK;
}
