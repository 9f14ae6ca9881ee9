// In-situ kernel timing (measurement only; off by default): selected launches are bracketed with hipEvents recorded on the SAME stream the kernel
// is launched on; maed_prof_collect() synchronises the events and returns total ms and launch counts per tag, maed_prof_flops() the algorithmic
// FLOPs the tagged launches declared, maed_prof_records() duration / FLOPs / algorithmic bytes of every launch of one tag.  bench.py uses this for the roofline numbers.  Storage and entry points: csrc/block.hip.
#pragma once
#include "common.cuh"

enum { PROF_ATTN_SP_FWD = 0, PROF_ATTN_TM_FWD, PROF_GEMM_QKV, PROF_GEMM_FC1, PROF_GEMM_FC2, PROF_ATTN_SP_BWD, PROF_ATTN_TM_BWD, PROF_GEMM_WGRAD,
       PROF_GEMM_PROJ, PROF_GEMM_DGRAD /* the four input-gradient GEMMs of a block */, PROF_LAYERNORM /* fwd + bwd */,
       PROF_TN_ALL /* EVERY maed_gemm_tn_wgrad launch: STE and backbone 1x1 convolutions */, PROF_TN_CONV /* every maed_conv3x3_wgrad launch */,
       PROF_ST_FWD, PROF_ST_BWD /* the fused attentive addition (maed_st_fused_fwd / _bwd) */, PROF_NTAGS };

bool maed_prof_on();
void maed_prof_open(int tag, hipStream_t s, hipEvent_t* a);
void maed_prof_close(int tag, hipStream_t s, hipEvent_t a, double flops, double bytes);

struct ProfScope {
    int tag; hipStream_t s; hipEvent_t a; bool on; double flops, bytes;      // bytes: algorithmic HBM bytes of the launch (operands once, results once)
    ProfScope(int tag_, void* stream, double flops_ = 0.0, double bytes_ = 0.0)
        : tag(tag_), s((hipStream_t)stream), a(nullptr), on(maed_prof_on()), flops(flops_), bytes(bytes_) {
        if (on) maed_prof_open(tag, s, &a);
    }
    ~ProfScope() { if (on) maed_prof_close(tag, s, a, flops, bytes); }
};
