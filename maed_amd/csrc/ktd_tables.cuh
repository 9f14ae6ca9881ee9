// Constant tables shared by the decoder-tail kernels (smpl.hip, tail_bwd.hip).
#pragma once
#define NJ 24
#define NV 6890
#define KTD_OUT 157   /* 24*6 pose + 10 shape + 3 cam: rows of the packed head GEMM */

// lib/models/ktd.py:10-35 ANCESTOR_INDEX, flattened: joint j has c_anc_cnt[j] ancestors c_anc[c_anc_start[j] ...];
// its ancestor weight block (6, 6*cnt) starts at 36*c_anc_start[j] inside the packed w_anc (3420 floats)
static __constant__ int c_anc_cnt[NJ] = {0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7, 7, 8, 8};
static __constant__ int c_anc_start[NJ + 1] = {0, 0, 1, 2, 3, 5, 7, 9, 12, 15, 18, 22, 26, 30, 34, 38, 43, 48, 53, 59, 65, 72, 79, 87, 95};
static __constant__ int c_anc[95] = {
    0, 0, 0, 0, 1, 0, 2, 0, 3, 0, 1, 4, 0, 2, 5, 0, 3, 6, 0, 1, 4, 7, 0, 2, 5, 8, 0, 3, 6, 9, 0, 3, 6, 9, 0, 3, 6, 9,
    0, 3, 6, 9, 12, 0, 3, 6, 9, 13, 0, 3, 6, 9, 14, 0, 3, 6, 9, 13, 16, 0, 3, 6, 9, 14, 17,
    0, 3, 6, 9, 13, 16, 18, 0, 3, 6, 9, 14, 17, 19, 0, 3, 6, 9, 13, 16, 18, 20, 0, 3, 6, 9, 14, 17, 19, 21};
