// One STE Block (vision_transformer.py:244-261, Attention st_mode='parallel' :146-158,176, Mlp :106-112)
// as a single host call: the whole kernel sequence of the block is enqueued from C++ on the caller's
// stream, so the Python autograd layer pays one ctypes call per block and direction instead of ~15-30
// op dispatches.  The residual stream stays fp32; activations are in the compute dtype.
//
// forward : LN1 -> qkv GEMM -> temporal attn + spatial attn (both read qkv in place) -> token means ->
//           ts_attn GEMM (F x 2C) -> attentive mix -> proj GEMM (+residual) -> LN2 -> fc1 GEMM (+GELU) ->
//           fc2 GEMM (+residual)
// backward: the mirror image; weight gradients are NT GEMMs on transposed copies (split-K, fp32 atomics
//           into the caller's gradient arena), bias gradients fall out of the transposes.
#include "common.cuh"
#include "gemm_x3.h"
#include "prof.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

// ---- in-situ kernel timing (csrc/prof.h): storage and entry points ----------------------------------------------
static bool g_prof = false;
struct ProfRec { hipEvent_t a, b; double flops, bytes; };
static std::vector<ProfRec> g_prof_ev[PROF_NTAGS];
bool maed_prof_on() { return g_prof; }
void maed_prof_open(int tag, hipStream_t s, hipEvent_t* a) { (void)tag; (void)hipEventCreate(a); (void)hipEventRecord(*a, s); }
void maed_prof_close(int tag, hipStream_t s, hipEvent_t a, double flops, double bytes) {
    hipEvent_t b;
    (void)hipEventCreate(&b); (void)hipEventRecord(b, s);
    g_prof_ev[tag].push_back(ProfRec{a, b, flops, bytes});
}
extern "C" int maed_prof_enable(int on) { g_prof = on != 0; return MAED_OK; }
extern "C" int maed_prof_ntags(void) { return PROF_NTAGS; }
extern "C" int maed_prof_flops(double* flops) {              // FLOPs declared by the tagged launches since the last collect (call BEFORE maed_prof_collect)
    for (int t = 0; t < PROF_NTAGS; ++t) {
        double tot = 0.0;
        for (auto& r : g_prof_ev[t]) tot += r.flops;
        if (flops) flops[t] = tot;
    }
    return MAED_OK;
}
// per-launch records of one tag, in launch order (call BEFORE maed_prof_collect): duration, declared FLOPs and algorithmic bytes; returns the number of records
// the tag holds (at most `cap` are written)
extern "C" int maed_prof_records(int tag, double* us, double* flops, double* bytes, int cap) {
    if (tag < 0 || tag >= PROF_NTAGS) return 0;
    int i = 0;
    for (auto& r : g_prof_ev[tag]) {
        if (i < cap) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            if (us) us[i] = 1e3 * (double)ms;
            if (flops) flops[i] = r.flops;
            if (bytes) bytes[i] = r.bytes;
        }
        ++i;
    }
    return i;
}
extern "C" int maed_prof_collect(double* ms_total, int* count) {
    for (int t = 0; t < PROF_NTAGS; ++t) {
        double tot = 0.0;
        for (auto& r : g_prof_ev[t]) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            tot += ms;
            (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        }
        if (ms_total) ms_total[t] = tot;
        if (count) count[t] = (int)g_prof_ev[t].size();
        g_prof_ev[t].clear();
    }
    return MAED_OK;
}
#define PROF(tag, expr) do { ProfScope ps__(tag, stream); MAED_PROPAGATE(expr); } while (0)

// ---- side stream for the weight-gradient GEMMs of the backward ---------------------------------------------------------------------
// dW += Y^T X depends only on operands the input-gradient chain has already produced and nothing downstream in the block reads dW, so the
// five TN GEMMs of a block run on a second stream beside the chain (input-gradient GEMMs, LayerNorm / mix / attention backward): both kinds
// of kernel leave most of the MFMA pipe idle on their own (0.2 / 0.3 of peak) and each kernel's ramp-up and tail fill with the other's
// workgroups.  Fences: an event per operand hand-over (main -> side), one before the dqkv buffer is re-used and one at the end of the block
// (side -> main), so everything after maed_ste_block_bwd on the caller's stream -- gradient all-reduce, Adam -- is ordered after the weight
// gradients.  The only objects the library ever creates besides the opt-in communicator: two non-blocking streams and two rings of timing-less events -- ALL of
// them made in ONE place, maed_init_runtime() (called by maed_init; a host that skipped maed_init gets them on first use through the same function), never
// destroyed.  MAED_WGRAD_SIDE_STREAM=0, the in-situ profiler (maed_prof_enable) and the f32 parity mode keep everything on the caller's stream.
struct SideStream {
    hipStream_t s = nullptr;
    hipStream_t s2 = nullptr;      // second side stream: the twin forward's cast pass (it feeds only the backward: it runs beside the NEXT block's forward)
    hipEvent_t ev[64];
    hipEvent_t cast_ev[2];         // "the cast that read work buffer k is done"; cast_pending: recorded and not yet waited for by a backward
    bool cast_pending[2] = {false, false};
    int next = 0;
    bool ok = false;
    // everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
    void fence(hipStream_t from, hipStream_t to) {
        hipEvent_t e = ev[next]; next = (next + 1) % 62;               // slots 62 / 63 are named fences, outside the ring
        (void)hipEventRecord(e, from);            // (a failure surfaces as the launch error of the kernels that follow)
        (void)hipStreamWaitEvent(to, e, 0);
    }
};
// every stream / event the library owns (this block driver's side streams + fence events, maed_stream_fence's ring): created here and nowhere else
static SideStream g_side;
static hipEvent_t g_fence_ring[64];
static bool g_fence_ok = false;
static std::once_flag g_runtime_once;
int maed_sk_init(void);
int maed_init_runtime(void) {
    std::call_once(g_runtime_once, [] {
        SideStream& ss = g_side;
        bool ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&ss.s2, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < 64; ++i) ok = hipEventCreateWithFlags(&ss.ev[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < 2; ++i) ok = hipEventCreateWithFlags(&ss.cast_ev[i], hipEventDisableTiming) == hipSuccess;
        ss.ok = ok;
        g_fence_ok = true;
        for (int i = 0; i < 64; ++i) if (hipEventCreateWithFlags(&g_fence_ring[i], hipEventDisableTiming) != hipSuccess) { g_fence_ok = false; break; }
        if (!ok || !g_fence_ok) (void)hipGetLastError();
        (void)maed_sk_init();        // slabs + flags of the persistent K-stream GEMM's hand-offs (csrc/gemm_sk.hip): the one device allocation the library owns
    });
    return (g_side.ok && g_fence_ok) ? MAED_OK : MAED_ERR_LAUNCH;
}
static SideStream* side_stream() {
    if (!maed_opt(MAED_OPT_SIDE_STREAM) || g_prof) return nullptr;
    (void)maed_init_runtime();
    return g_side.ok ? &g_side : nullptr;
}

// "everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on": one event record + one stream wait from a ring of timing-less
// events the library owns.  For hosts that run single launches on a second stream of their own (maed_amd/ops.py: the backbone's weight-gradient GEMMs): the same
// fence through the framework costs a Python-level event object, a record and a wait per use.
// The cast passes of twin forwards run on a library stream and write the CALLER's arenas (maed_ste_block_fwd_twin): a backward joins them by itself, a host that
// drops the graph without a backward -- or frees / re-sizes a work buffer -- joins them here first: `stream` waits for every cast still pending.
extern "C" int maed_ste_block_twin_join(void* stream) {
    (void)maed_init_runtime();
    SideStream& ss = g_side;
    if (!ss.ok) return MAED_OK;
    for (int k = 0; k < 2; ++k)
        if (ss.cast_pending[k]) { MAED_HIP(hipStreamWaitEvent((hipStream_t)stream, ss.cast_ev[k], 0), "ste_block_twin_join: stream wait"); ss.cast_pending[k] = false; }
    return MAED_OK;
}

extern "C" int maed_stream_fence(void* from_stream, void* to_stream) {
    static std::atomic<unsigned> next{0};
    (void)maed_init_runtime();
    MAED_CHECK_ARG(g_fence_ok, MAED_ERR_LAUNCH, "stream_fence: event ring could not be created");
    if (from_stream == to_stream) return MAED_OK;
    hipEvent_t e = g_fence_ring[next.fetch_add(1, std::memory_order_relaxed) & 63];
    MAED_HIP(hipEventRecord(e, (hipStream_t)from_stream), "stream_fence: record");
    MAED_HIP(hipStreamWaitEvent((hipStream_t)to_stream, e, 0), "stream_fence: wait");
    return MAED_OK;
}

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct SavedLayout {
    size_t ln1, mean1, rstd1, qkv, xs, xt, lse_s, lse_t, means, logits, mix, xmid, mean2, rstd2, ln2, hpre, hact, st_sync, st_ex, total;
};
struct ScratchLayout {
    size_t dyc, dyt, bigA, bigT, xT, act, dxs, dxt, dxmid, dlog, dmeans, dlogT, meansT, ws, total;
    int64_t Mp, Fp;
};

SavedLayout saved_layout(const maed_block_dims& d) {
    const size_t es = dtype_size(d.dtype);
    const size_t M = (size_t)d.F * d.P, C = d.C, Hd = d.hidden;
    SavedLayout s{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    s.ln1 = take(M * C * es); s.mean1 = take(M * 4); s.rstd1 = take(M * 4);
    s.qkv = take(M * 3 * C * es); s.xs = take(M * C * es); s.xt = take(M * C * es);
    s.lse_s = take((size_t)d.F * d.H * d.P * 4); s.lse_t = take((size_t)d.F * d.H * d.P * 4);
    s.means = take((size_t)d.F * 2 * C * es); s.logits = take((size_t)d.F * 2 * C * 4);
    s.mix = take(M * C * es); s.xmid = take(M * C * 4);
    s.mean2 = take(M * 4); s.rstd2 = take(M * 4); s.ln2 = take(M * C * es);
    s.hpre = take(M * Hd * es); s.hact = take(M * Hd * es);
    s.st_sync = take((size_t)d.F * 16 * 4); s.st_ex = take((size_t)d.F * 2 * C * 4);      // scratch of the fused attentive addition (maed_st_fused_fwd / _bwd)
    s.total = o;
    return s;
}

ScratchLayout scratch_layout(const maed_block_dims& d) {
    const size_t es = dtype_size(d.dtype);
    const size_t M = (size_t)d.F * d.P, C = d.C, Hd = d.hidden;
    ScratchLayout s{};
    s.Mp = (int64_t)((M + 63) / 64 * 64);
    s.Fp = (int64_t)(((size_t)d.F + 63) / 64 * 64);
    const size_t big = Hd > 3 * C ? Hd : 3 * C;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
    s.dyc = take(M * C * es); s.dyt = take(C * (size_t)s.Mp * es);
    s.bigA = take(M * big * es); s.bigT = take(big * (size_t)s.Mp * es);
    s.xT = take(C * (size_t)s.Mp * es); s.act = take(M * C * es);
    s.dxs = take(M * C * es); s.dxt = take(M * C * es); s.dxmid = take(M * C * 4);
    s.dlog = take((size_t)d.F * 2 * C * es); s.dmeans = take((size_t)d.F * 2 * C * es);
    s.dlogT = take(2 * C * (size_t)s.Fp * es); s.meansT = take(2 * C * (size_t)s.Fp * es);
    s.ws = take((size_t)d.F * 2 * C * 4);
    s.total = o;
    return s;
}

}  // namespace
// layernorm.hip: dgamma/dbeta through per-workgroup partials + a column sum instead of contended atomics
size_t maed_layernorm_bwd_partials_bytes(int64_t rows, int C);
int maed_layernorm_bwd_ws(const void* dy, int dtype, const float* x, int64_t x_row_stride, const float* gamma, const float* mean, const float* rstd,
                          const float* dres_in, float* dx_out, void* dx_twin, float* dgamma, float* dbeta, int64_t rows, int C, float* partials,
                          void* stream, bool finish = true, uint32_t* clear = nullptr, int clear_words = 0);
int maed_layernorm_affine_finish(const float* partials, int64_t rows, int C, float* dgamma, float* dbeta, void* stream);
int maed_layernorm_fwd_ws(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, void* y, int dtype, float* mean, float* rstd, int64_t rows,
                          int C, float eps, uint32_t* clear, int clear_words, void* stream, void* y_lo = nullptr);
int maed_st_fused_fwd_ws(const void* x_s, const void* x_t, const void* w_ts, const float* b_ts, void* means, float* logits, void* mix,
                         uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream);
int maed_st_fused_bwd_ws(const void* dmix, const void* x_s, const void* x_t, const float* logits, const void* wt_ts, void* dlogits, void* dx_s,
                         void* dx_t, uint32_t* sync, float* ex, int F, int P, int C, int dtype, bool clear_sync, uint32_t arrive_base, void* stream);
namespace {

int check_dims(const maed_block_dims* d, const char* who) {
    MAED_CHECK_ARG(d, MAED_ERR_ARG, "%s: null dims", who);
    MAED_CHECK_ARG(d->dtype == MAED_F32 || d->dtype == MAED_BF16, MAED_ERR_ARG, "%s: bad dtype %d", who, d->dtype);
    MAED_CHECK_ARG(d->F > 0 && d->P > 0 && d->H > 0 && d->T > 0 && d->hidden > 0, MAED_ERR_SHAPE, "%s: non-positive extent", who);
    MAED_CHECK_ARG(d->C == d->H * HEAD_DIM, MAED_ERR_SHAPE, "%s: C=%d must equal 64*H (H=%d)", who, d->C, d->H);
    MAED_CHECK_ARG(d->F % d->T == 0, MAED_ERR_SHAPE, "%s: F=%d not a multiple of T=%d", who, d->F, d->T);
    MAED_CHECK_ARG(d->hidden % 64 == 0, MAED_ERR_SHAPE, "%s: hidden=%d must be a multiple of 64", who, d->hidden);
    return MAED_OK;
}

// the attentive addition as one launch per direction (bf16 MFMA mode; MAED_OPT_ST_FUSED = 0 keeps the four-launch sequence: A/B knob)
bool st_fused(const maed_block_dims& d) {
    // (P <= 224: the 7-chunk instantiations, two workgroups per CU; at cfg5's P = 257 / C = 768 the 9-chunk backward runs one workgroup per CU and measured 0.3 ms
    //  per step SLOWER than the four-launch sequence -- profiles/r04_st_fused_bench_ab.txt)
    return d.impl != MAED_IMPL_VALU && maed_opt(MAED_OPT_ST_FUSED) && d.P <= 224 && maed_st_fused_supported(d.P, d.C, d.dtype);
}

// split-K so that a weight-gradient GEMM (small output, huge K) still fills 256 CUs
int pick_splitk(int64_t Mo, int64_t No, int64_t K, int dtype) {
    const int64_t tile = (dtype == MAED_BF16) ? 128 : 64;
    const int64_t kt = (dtype == MAED_BF16) ? 64 : 16;
    const int64_t tiles = ((Mo + tile - 1) / tile) * ((No + tile - 1) / tile);
    int64_t s = (1024 + tiles - 1) / tiles;
    const int64_t maxs = K / (kt * 4) > 0 ? K / (kt * 4) : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return (int)s;
}

// dW[No,Ko] += Yt[No,Mp] * Xt[Ko,Mp]^T
int wgrad(const void* Yt, const void* Xt, int64_t No, int64_t Ko, int64_t Mp, float* dW, const maed_block_dims& d, void* st) {
    return maed_gemm_nt(Yt, Mp, Xt, Mp, No, Ko, Mp, d.dtype, MAED_EPI_ATOMIC_F32, nullptr, dW, Ko, nullptr, nullptr, 0,
                        pick_splitk(No, Ko, Mp, d.dtype), d.impl == MAED_IMPL_VALU ? MAED_IMPL_VALU : MAED_IMPL_AUTO, st);
}

}  // namespace

extern "C" size_t maed_ste_block_saved_bytes(const maed_block_dims* d) { return d ? saved_layout(*d).total : 0; }
extern "C" size_t maed_ste_block_scratch_bytes(const maed_block_dims* d) { return d ? scratch_layout(*d).total : 0; }

// where the forward writes what it saves: one pointer per field of SavedLayout.  Plain forward: every field inside `saved`.  Twin forward (maed_ste_block_fwd_twin):
// the fields in the compute dtype go to an fp32 work buffer (they are the forward chain's operands), the fp32 fields straight into the bf16-layout arena the
// backward will read.
struct FwdBufs { char *ln1, *mean1, *rstd1, *qkv, *xs, *xt, *lse_s, *lse_t, *means, *logits, *mix, *xmid, *mean2, *rstd2, *ln2, *hpre, *hact, *st_sync, *st_ex;
                 // twin forward only: bf16 twins written by the producing GEMM's epilogue (no cast pass for them); hpre then holds bf16 (nothing of the forward reads it)
                 char *tw_qkv = nullptr, *tw_hact = nullptr; bool hpre_bf16 = false;
                 // ... with plane storage of fc1's activation (MAED_OPT_X3_PLANES): its lo plane and the planes of fc2's weight, in the work buffer's (unused) fp32 field
                 char *hact_lo = nullptr, *wfc2_hi = nullptr, *wfc2_lo = nullptr; int planes_variant = 0;
                 // ... and of the two LayerNorm outputs (MAED_OPT_X3_PLANES_LN): hi planes straight into the bf16 arena (they ARE the twins), lo planes + the planes of
                 // the qkv / fc1 weights in the work buffer; qkv and fc1 then run on the plane kernel too
                 char *ln1_hi = nullptr, *ln1_lo = nullptr, *ln2_hi = nullptr, *ln2_lo = nullptr, *wqkv_hi = nullptr, *wqkv_lo = nullptr, *wfc1_hi = nullptr, *wfc1_lo = nullptr; };
static FwdBufs fwd_bufs(char* act_base, const SavedLayout& A, char* f32_base, const SavedLayout& Fl) {
    FwdBufs b;
    b.ln1 = act_base + A.ln1; b.qkv = act_base + A.qkv; b.xs = act_base + A.xs; b.xt = act_base + A.xt; b.means = act_base + A.means; b.mix = act_base + A.mix;
    b.ln2 = act_base + A.ln2; b.hpre = act_base + A.hpre; b.hact = act_base + A.hact;
    b.mean1 = f32_base + Fl.mean1; b.rstd1 = f32_base + Fl.rstd1; b.lse_s = f32_base + Fl.lse_s; b.lse_t = f32_base + Fl.lse_t; b.logits = f32_base + Fl.logits;
    b.xmid = f32_base + Fl.xmid; b.mean2 = f32_base + Fl.mean2; b.rstd2 = f32_base + Fl.rstd2; b.st_sync = f32_base + Fl.st_sync; b.st_ex = f32_base + Fl.st_ex;
    return b;
}

int maed_gemm_nt_twin(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int dtype, int epilogue, const float* bias, void* out,
                      int64_t ldo, void* out2, const void* aux, int64_t ldaux, int splitk, int impl, void* stream, void* twin, bool out2_bf16, void* lo);      // gemm.hip
static int block_fwd_bufs(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, const FwdBufs& B, bool for_backward, void* stream) {
    const int64_t M = (int64_t)d->F * d->P;
    const int C = d->C, Hd = d->hidden, dt = d->dtype;
    const int gi = d->impl == MAED_IMPL_VALU ? MAED_IMPL_VALU : MAED_IMPL_AUTO;
    const float scale = 1.0f / sqrtf((float)HEAD_DIM);
    float* logits = (float*)B.logits;

    // (the LayerNorm kernel also zeroes the per-frame arrival counters of the fused attentive addition further down: no memset launch of their own)
    const bool stf = st_fused(*d);
    const bool piggy = stf && maed_opt(MAED_OPT_ST_FUSED) != 2 && (int64_t)d->F * 16 <= 256 * ((M + 31) / 32);       // (the backward LayerNorm's grid is the smaller one; otherwise the fused call clears itself; option value 2: A/B knob -- memset nodes, LayerNorm column sums on the caller's stream)
    if (B.ln1_lo) {
        // weights of the block's plane products, split once per call: [fc2 | qkv | fc1]
        const float* wsrc[3] = {(const float*)p->w_fc2, (const float*)p->w_qkv, (const float*)p->w_fc1};
        void* wh[3] = {B.wfc2_hi, B.wqkv_hi, B.wfc1_hi}; void* wl[3] = {B.wfc2_lo, B.wqkv_lo, B.wfc1_lo};
        const int64_t wn[3] = {(int64_t)C * Hd, (int64_t)3 * C * C, (int64_t)C * Hd};
        MAED_PROPAGATE(maed_split_planes_launch(3, wsrc, wh, wl, wn, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("ste_block_fwd_twin: weight planes");
        MAED_PROPAGATE(maed_layernorm_fwd_ws(x_in, C, p->ln1_g, p->ln1_b, B.ln1_hi, MAED_BF16, (float*)(B.mean1), (float*)(B.rstd1), M, C, d->eps,
                                             piggy ? (uint32_t*)(B.st_sync) : nullptr, piggy ? d->F * 16 : 0, stream, B.ln1_lo));
        EpiArgs eq{};
        eq.bias = p->b_qkv; eq.out = B.qkv; eq.ldo = 3 * C; eq.twin = B.tw_qkv;
        PROF(PROF_GEMM_QKV, maed_gemm_nt_x3p_launch(MAED_EPI_STORE, 2, B.ln1_hi, B.ln1_lo, C, B.wqkv_hi, B.wqkv_lo, C, M, 3 * C, C, eq, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("ste_block_fwd_twin: qkv on planes");
    } else {
    MAED_PROPAGATE(maed_layernorm_fwd_ws(x_in, C, p->ln1_g, p->ln1_b, B.ln1, dt, (float*)(B.mean1), (float*)(B.rstd1), M, C, d->eps,
                                         piggy ? (uint32_t*)(B.st_sync) : nullptr, piggy ? d->F * 16 : 0, stream));
    PROF(PROF_GEMM_QKV, maed_gemm_nt_twin(B.ln1, C, p->w_qkv, C, M, 3 * C, C, dt, MAED_EPI_STORE, p->b_qkv, B.qkv, 3 * C, nullptr, nullptr, 0, 1, gi, stream, B.tw_qkv, false, nullptr));
    }
    {   // the two attention branches read the same qkv and write disjoint outputs: temporal on the side stream beside spatial
        SideStream* ss = (d->impl != MAED_IMPL_VALU && (dt == MAED_BF16 || maed_x3_planes())) ? side_stream() : nullptr;
        void* tst = ss ? (void*)ss->s : stream;
        if (ss) ss->fence((hipStream_t)stream, ss->s);
        { ProfScope ps__(PROF_ATTN_TM_FWD, tst); MAED_PROPAGATE(maed_attn_temporal_fwd(B.qkv, B.xt, (float*)(B.lse_t), d->F, d->P, d->H, d->T, scale, dt, tst)); }
        PROF(PROF_ATTN_SP_FWD, maed_attn_spatial_fwd(B.qkv, B.xs, (float*)(B.lse_s), d->F, d->P, d->H, scale, dt, d->impl, stream));
        if (ss) ss->fence(ss->s, (hipStream_t)stream);
    }
    if (stf) {       // token means + ts_attn Linear + pair softmax + mix: one launch, x_s / x_t read once (elementwise.hip)
        PROF(PROF_ST_FWD, maed_st_fused_fwd_ws(B.xs, B.xt, p->w_ts, p->b_ts, B.means, logits, B.mix, (uint32_t*)(B.st_sync), (float*)(B.st_ex),
                                               d->F, d->P, C, dt, !piggy, 0u, stream));
    } else {
        MAED_PROPAGATE(maed_st_colmean(B.xs, B.xt, B.means, logits /* scratch, overwritten below */, d->F, d->P, C, dt, stream));
        MAED_PROPAGATE(maed_gemm_nt(B.means, 2 * C, p->w_ts, 2 * C, d->F, 2 * C, 2 * C, dt, MAED_EPI_STORE_F32, p->b_ts, logits, 2 * C, nullptr, nullptr, 0, 1, gi, stream));
        MAED_PROPAGATE(maed_st_mix_fwd(B.xs, B.xt, logits, B.mix, d->F, d->P, C, dt, stream));
    }
    PROF(PROF_GEMM_PROJ, maed_gemm_nt(B.mix, C, p->w_proj, C, M, C, C, dt, MAED_EPI_RESID_F32, p->b_proj, B.xmid, C, nullptr, x_in, C, 1, gi, stream));
    if (B.ln2_lo) MAED_PROPAGATE(maed_layernorm_fwd_ws((const float*)(B.xmid), C, p->ln2_g, p->ln2_b, B.ln2_hi, MAED_BF16, (float*)(B.mean2), (float*)(B.rstd2), M, C, d->eps,
                                                       nullptr, 0, stream, B.ln2_lo));
    else MAED_PROPAGATE(maed_layernorm_fwd((const float*)(B.xmid), C, p->ln2_g, p->ln2_b, B.ln2, dt, (float*)(B.mean2), (float*)(B.rstd2), M, C, d->eps, stream));
    if (B.hact_lo) {
        // twin forward with plane storage (MAED_OPT_X3_PLANES): fc1's epilogue leaves the activation as (hi, lo) bf16 planes only -- hi IS the backward's twin, 4 bytes
        // per element instead of fp32 + twin = 6 -- and fc2 multiplies the planes (csrc/gemm_x3p.hip: the same three products per K step, bit for bit); its weight is
        // split once per call (1 M elements)
        if (!B.ln1_lo) {
            const float* wsrc[1] = {(const float*)p->w_fc2}; void* wh[1] = {B.wfc2_hi}; void* wl[1] = {B.wfc2_lo}; const int64_t wn[1] = {(int64_t)C * Hd};
            MAED_PROPAGATE(maed_split_planes_launch(1, wsrc, wh, wl, wn, (hipStream_t)stream));
            MAED_CHECK_LAUNCH("ste_block_fwd_twin: weight planes");
        }
        if (B.ln2_lo) {
            EpiArgs e1{};
            e1.bias = p->b_fc1; e1.ldo = Hd; e1.out2 = B.hpre; e1.out2_bf16 = true; e1.twin = B.tw_hact; e1.lo = B.hact_lo;
            PROF(PROF_GEMM_FC1, maed_gemm_nt_x3p_launch(MAED_EPI_GELU, 2, B.ln2_hi, B.ln2_lo, C, B.wfc1_hi, B.wfc1_lo, C, M, Hd, C, e1, (hipStream_t)stream));
            MAED_CHECK_LAUNCH("ste_block_fwd_twin: fc1 on planes");
        } else
        PROF(PROF_GEMM_FC1, maed_gemm_nt_twin(B.ln2, C, p->w_fc1, C, M, Hd, C, dt, MAED_EPI_GELU, p->b_fc1, nullptr, Hd, B.hpre, nullptr, 0, 1, gi, stream, B.tw_hact, true, B.hact_lo));
        EpiArgs e2{};
        e2.bias = p->b_fc2; e2.out = x_out; e2.ldo = C; e2.aux = B.xmid; e2.ldaux = C;
        PROF(PROF_GEMM_FC2, maed_gemm_nt_x3p_launch(MAED_EPI_RESID_F32, B.planes_variant, B.tw_hact, B.hact_lo, Hd, B.wfc2_hi, B.wfc2_lo, Hd, M, C, Hd, e2, (hipStream_t)stream));
        MAED_CHECK_LAUNCH("ste_block_fwd_twin: fc2 on planes");
        return MAED_OK;
    }
    PROF(PROF_GEMM_FC1, maed_gemm_nt_twin(B.ln2, C, p->w_fc1, C, M, Hd, C, dt, MAED_EPI_GELU, p->b_fc1, B.hact, Hd, for_backward ? B.hpre : nullptr /* only GELU' reads it */, nullptr, 0, 1, gi,
                                          stream, B.tw_hact, B.hpre_bf16, nullptr));
    PROF(PROF_GEMM_FC2, maed_gemm_nt(B.hact, Hd, p->w_fc2, Hd, M, C, Hd, dt, MAED_EPI_RESID_F32, p->b_fc2, x_out, C, nullptr, B.xmid, C, 1, gi, stream));
    return MAED_OK;
}

static int block_fwd(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, void* saved, bool for_backward, void* stream) {
    MAED_PROPAGATE(check_dims(d, "ste_block_fwd"));
    MAED_CHECK_ARG(p && x_in && x_out && saved, MAED_ERR_ARG, "ste_block_fwd: null pointer");
    MAED_CHECK_ARG(is_aligned(saved, 256), MAED_ERR_ALIGN, "ste_block_fwd: saved buffer must be 256-B aligned");
    const SavedLayout L = saved_layout(*d);
    return block_fwd_bufs(d, p, x_in, x_out, fwd_bufs((char*)saved, L, (char*)saved, L), for_backward, stream);
}

// ---- "bf16x3 forward / bf16 backward from bf16 twins" (round 5) -----------------------------------------------------------------------------------------
// The accurate mode's forward needs fp32 operands (split-bf16 products: north_star's 1e-3 on the outputs); its backward does not -- gradients of the bf16 mode's
// quality are what the bf16 headline trains with.  So the forward runs on fp32 tensors in a transient work buffer, and what the backward will read is saved as bf16
// TWINS in the bf16 mode's arena layout: the backward IS the bf16 mode's (maed_ste_block_bwd with dtype MAED_BF16 and bf16 weight images), at its speed, instead of
// one-plane products on fp32-stored operands (MAED_F32X1: bound by fp32 operand traffic and conversion, 9.2 vs 4.8 ms per step at cfg3).
struct CastTab { const float* src[12]; bf16* dst[12]; long long n8[12]; };
__global__ __launch_bounds__(256) void cast_table_kernel(CastTab t) {
    const float* __restrict__ src = t.src[blockIdx.y];
    bf16* __restrict__ dst = t.dst[blockIdx.y];
    const long long n8 = t.n8[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float v[8];
        ld8(src + i * 8, v);
        st8_nt(dst + i * 8, v);
    }
}

extern "C" size_t maed_ste_block_twin_work_bytes(const maed_block_dims* d) {
    if (!d) return 0;
    maed_block_dims d32 = *d; d32.dtype = MAED_F32;
    return saved_layout(d32).total + ((size_t)2 * d->C * d->hidden + (size_t)3 * d->C * d->C) * 4;       // + the bf16 planes of the fc2, qkv and fc1 weights (MAED_OPT_X3_PLANES / _LN)
}

extern "C" int maed_ste_block_fwd_twin(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, void* saved_bf16, void* work_f32,
                                       int work_slot, void* stream) {
    MAED_PROPAGATE(check_dims(d, "ste_block_fwd_twin"));
    MAED_CHECK_ARG(d->dtype == MAED_F32, MAED_ERR_ARG, "ste_block_fwd_twin: dims describe the FORWARD (dtype MAED_F32); the arena is laid out for MAED_BF16");
    MAED_CHECK_ARG(p && x_in && x_out && saved_bf16 && work_f32, MAED_ERR_ARG, "ste_block_fwd_twin: null pointer");
    MAED_CHECK_ARG(is_aligned(saved_bf16, 256) && is_aligned(work_f32, 256), MAED_ERR_ALIGN, "ste_block_fwd_twin: buffers must be 256-B aligned");
    MAED_CHECK_ARG(d->C % 8 == 0 && d->hidden % 8 == 0, MAED_ERR_SHAPE, "ste_block_fwd_twin: C and hidden must be multiples of 8");
    MAED_CHECK_ARG(work_slot == 0 || work_slot == 1, MAED_ERR_ARG, "ste_block_fwd_twin: work_slot must be 0 or 1");
    maed_block_dims d16 = *d; d16.dtype = MAED_BF16;
    const SavedLayout L32 = saved_layout(*d), L16 = saved_layout(d16);
    char* w = (char*)work_f32; char* sv = (char*)saved_bf16;
    // the cast pass runs on a side stream of its own, beside the next block's forward (which writes the OTHER work buffer: the caller alternates work_slot 0 / 1 and
    // hands each slot its own buffer); before this forward overwrites its buffer, the cast that last read it must be done
    SideStream* ss = side_stream();
    if (ss && ss->cast_pending[work_slot]) MAED_HIP(hipStreamWaitEvent((hipStream_t)stream, ss->cast_ev[work_slot], 0), "ste_block_fwd_twin: stream wait");
    // the three largest fields -- qkv and fc1's pre- / post-activation, 11 of the block's 16 M x C units -- get their twins from the producing GEMM's epilogue
    // (the pre-activation as bf16 only: the forward never reads it); the cast pass handles the rest
    FwdBufs fb = fwd_bufs(w, L32, sv, L16);
    fb.tw_qkv = sv + L16.qkv; fb.tw_hact = sv + L16.hact; fb.hpre = sv + L16.hpre; fb.hpre_bf16 = true;
    const int pv = maed_opt(MAED_OPT_X3_PLANES);
    char* const wpl = w + L32.total;                                       // (fields are 256-byte aligned, so is the total)
    if (pv && maed_x3_planes() == 2 && d->impl != MAED_IMPL_VALU
        && maed_x3p_shape_ok(fb.tw_hact, w + L32.hact, d->hidden, wpl, wpl + (size_t)d->C * d->hidden * 2, d->hidden, (int64_t)d->F * d->P, d->C, d->hidden)) {
        // the fp32 field of fc1's activation holds its lo plane (half of it); the planes of fc2's weight sit behind the fp32 layout
        fb.hact_lo = w + L32.hact; fb.wfc2_hi = wpl; fb.wfc2_lo = wpl + (size_t)d->C * d->hidden * 2; fb.planes_variant = pv;
        if (maed_opt(MAED_OPT_X3_PLANES_LN) && d->C % 32 == 0) {
            const size_t CH2 = (size_t)d->C * d->hidden * 2, CC2 = (size_t)3 * d->C * d->C * 2;
            fb.ln1_hi = sv + L16.ln1; fb.ln1_lo = w + L32.ln1; fb.ln2_hi = sv + L16.ln2; fb.ln2_lo = w + L32.ln2;
            fb.wqkv_hi = wpl + 2 * CH2; fb.wqkv_lo = fb.wqkv_hi + CC2; fb.wfc1_hi = fb.wqkv_lo + CC2; fb.wfc1_lo = fb.wfc1_hi + CH2;
        }
    }
    MAED_PROPAGATE(block_fwd_bufs(d, p, x_in, x_out, fb, true, stream));
    const long long M = (long long)d->F * d->P, C = d->C;
    CastTab t{};
    int k = 0;
    auto add = [&](size_t o32, size_t o16, long long n) { t.src[k] = (const float*)(w + o32); t.dst[k] = (bf16*)(sv + o16); t.n8[k] = n / 8; ++k; };
    if (!fb.ln1_lo) add(L32.ln1, L16.ln1, M * C);
    add(L32.xs, L16.xs, M * C); add(L32.xt, L16.xt, M * C);
    add(L32.means, L16.means, (long long)d->F * 2 * C); add(L32.mix, L16.mix, M * C);
    if (!fb.ln2_lo) add(L32.ln2, L16.ln2, M * C);
    hipStream_t cs = (hipStream_t)stream;
    if (ss) { ss->fence((hipStream_t)stream, ss->s2); cs = ss->s2; }
    hipLaunchKernelGGL(cast_table_kernel, dim3(1024, k), dim3(256), 0, cs, t);
    MAED_CHECK_LAUNCH("ste_block_fwd_twin: cast");
    if (ss) { MAED_HIP(hipEventRecord(ss->cast_ev[work_slot], ss->s2), "ste_block_fwd_twin: event"); ss->cast_pending[work_slot] = true; }
    return MAED_OK;
}

extern "C" int maed_ste_block_fwd(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, void* saved, void* stream) {
    return block_fwd(d, p, x_in, x_out, saved, true, stream);
}
// the same forward when no backward will follow (inference): what only the backward reads is not written -- fc1's pre-activation, 103 MB per block at cfg3
extern "C" int maed_ste_block_infer(const maed_block_dims* d, const maed_block_params* p, const float* x_in, float* x_out, void* work, void* stream) {
    return block_fwd(d, p, x_in, x_out, work, false, stream);
}

extern "C" int maed_ste_block_bwd(const maed_block_dims* d, const maed_block_params* p, const maed_block_grads* g,
                                  const float* x_in, const float* dx_out, float* dx_in, void* saved, void* scratch,
                                  const void* dx_out_twin, void* dx_in_twin, void* stream) {
    MAED_PROPAGATE(check_dims(d, "ste_block_bwd"));
    MAED_CHECK_ARG(p && g && x_in && dx_out && dx_in && saved && scratch, MAED_ERR_ARG, "ste_block_bwd: null pointer");
    MAED_CHECK_ARG(p->wt_qkv && p->wt_ts && p->wt_proj && p->wt_fc1 && p->wt_fc2, MAED_ERR_ARG, "ste_block_bwd: transposed weights missing");
    MAED_CHECK_ARG(is_aligned(saved, 256) && is_aligned(scratch, 256), MAED_ERR_ALIGN, "ste_block_bwd: saved/scratch must be 256-B aligned");
    const SavedLayout L = saved_layout(*d);
    const ScratchLayout S = scratch_layout(*d);
    char* sv = (char*)saved; char* sc = (char*)scratch;
    if (d->dtype == MAED_BF16) {        // twin forwards leave their cast passes on a side stream: the arena is complete once they are done
        SideStream* ssc = side_stream();
        for (int k = 0; ssc && k < 2; ++k)
            if (ssc->cast_pending[k]) { MAED_HIP(hipStreamWaitEvent((hipStream_t)stream, ssc->cast_ev[k], 0), "ste_block_bwd: stream wait"); ssc->cast_pending[k] = false; }
    }
    const int64_t M = (int64_t)d->F * d->P, Mp = S.Mp, Fp = S.Fp;
    const int C = d->C, Hd = d->hidden, dt = d->dtype;
    const int gi = d->impl == MAED_IMPL_VALU ? MAED_IMPL_VALU : MAED_IMPL_AUTO;
    const float scale = 1.0f / sqrtf((float)HEAD_DIM);
    const float* logits = (const float*)(sv + L.logits);
    float* dxmid = (float*)(sc + S.dxmid);

    // (f32: the split TN kernel needs N, K and the row strides -- C, 2C, 3C, hidden -- to be multiples of 4; C = 64 H always is, a hidden width that is not falls
    //  through to the exact transposed-copy path below, never less accurate than asked for, as maed_gemm_nt does)
    if (d->impl != MAED_IMPL_VALU && (dt == MAED_BF16 || (maed_x3_planes() && d->hidden % 4 == 0))) {
        // ---- bf16, and f32 in the split-bf16 matmul mode: weight gradients straight from the row-major operands (maed_gemm_tn_wgrad), no transposed copies ----
        const void* dyc = dt == MAED_F32 ? (const void*)dx_out : dx_out_twin;   // dx_out in the compute dtype
        if (!dyc) {                                                      // first block of the backward: cast once
            MAED_PROPAGATE(maed_transpose_cast(dx_out, MAED_F32, C, M, C, nullptr, 0, sc + S.dyc, C, nullptr, dt, stream));
            dyc = sc + S.dyc;
        }
        void* dxmid_tw = dt == MAED_F32 ? (void*)dxmid : (void*)(sc + S.dyt);   // compute-dtype twin of dx_mid (f32: dx_mid itself; bf16: the old transpose slot)
        SideStream* ss = side_stream();
        hipStream_t main_s = (hipStream_t)stream;
        void* wst = ss ? (void*)ss->s : stream;                          // where the weight-gradient GEMMs go
#define TO_SIDE() do { if (ss) ss->fence(main_s, ss->s); } while (0)   /* operands produced so far on the chain are visible to the side stream */
#define FROM_SIDE() do { if (ss) ss->fence(ss->s, main_s); } while (0) /* the chain waits for the weight gradients issued so far */
#define WGRAD(expr) do { ProfScope ps__(PROF_GEMM_WGRAD, wst); MAED_PROPAGATE(expr); } while (0)
        // matrix-product engine of the backward products on fp32 tensors: the process-wide one, or one bf16 plane (MAED_OPT_F32_BWD_X1: "bf16x3 forward / bf16 backward")
        const int dmm = (dt == MAED_F32 && maed_opt(MAED_OPT_F32_BWD_X1)) ? MAED_F32X1 : dt;
        // MLP: x_out = x_mid + fc2(gelu(fc1(ln2(x_mid))))
        // ALIASING (the header lets dx_in alias dx_out): in f32 the fc2 weight gradient reads dx_out ITSELF on the side stream, and dx_in is written only by the
        // closing layernorm_bwd_ws.  What orders the two: ev[63] is recorded on the side stream behind the fc1 weight gradient (hence behind this one) and the
        // caller's stream waits for it before the attention backward -- long before the closing LayerNorm.  Keep that wait if the order below is ever changed.
        TO_SIDE();
        WGRAD(maed_gemm_tn_wgrad(dyc, C, sv + L.hact, Hd, M, C, Hd, g->w_fc2, Hd, g->b_fc2, dmm, wst));
        PROF(PROF_GEMM_DGRAD, maed_gemm_nt(dyc, C, p->wt_fc2, C, M, Hd, C, dmm, MAED_EPI_MUL_DGELU, nullptr, sc + S.bigA, Hd, nullptr, sv + L.hpre, Hd, 1, gi, stream));
        TO_SIDE();
        WGRAD(maed_gemm_tn_wgrad(sc + S.bigA, Hd, sv + L.ln2, C, M, Hd, C, g->w_fc1, C, g->b_fc1, dmm, wst));
        if (ss) MAED_HIP(hipEventRecord(ss->ev[63], ss->s), "ste_block_bwd: event");                       // "fc1 weight gradient done" (named slot, outside the ring)
        PROF(PROF_GEMM_DGRAD, maed_gemm_nt(sc + S.bigA, Hd, p->wt_fc1, Hd, M, C, Hd, dmm, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));
        // LayerNorm dgamma/dbeta via partials in the (here unused) transpose slot instead of contended atomics (measured on MI355X,
        // profiles/r02_call2_steady_*.csv: 0.629 -> 0.503 + 0.080 ms per step)
        const size_t big_t_bytes = (size_t)(Hd > 3 * C ? Hd : 3 * C) * (size_t)Mp * dtype_size(dt);      // S.bigT: only the exact-f32 path transposes into it
        // round 4 experiment (MAED_OPT_ST_FUSED = 4): the closing column sums (12 launches of ~7 us per step) on the side stream -- nothing reads the LayerNorm
        // gradients before the end of the block (FROM_SIDE); LN2 and LN1 get their own partial regions, LN1's kernel would otherwise overwrite what the side stream
        // still sums.  Measured same-box: 21.07 / 21.20 ms against 21.04 / 21.10 with the sums on the caller's stream -- the two fences cost what the launches
        // did; not the default.
        const size_t ln_pb = (maed_layernorm_bwd_partials_bytes(M, C) + 255) & ~(size_t)255;
        float* ln_part = 2 * ln_pb <= big_t_bytes ? (float*)(sc + S.bigT) : nullptr;
        float* ln_part1 = ln_part ? (float*)(sc + S.bigT + ln_pb) : nullptr;
        const bool ln_side = ss != nullptr && ln_part != nullptr && maed_opt(MAED_OPT_ST_FUSED) == 4;
        const bool piggy = st_fused(*d) && maed_opt(MAED_OPT_ST_FUSED) != 2 && (int64_t)d->F * 16 <= 256 * ((M + 31) / 32);
        MAED_PROPAGATE(maed_layernorm_bwd_ws(sc + S.act, dt, (const float*)(sv + L.xmid), C, p->ln2_g, (const float*)(sv + L.mean2), (const float*)(sv + L.rstd2),
                                             dx_out, dxmid, dt == MAED_F32 ? nullptr : dxmid_tw, g->ln2_g, g->ln2_b, M, C, ln_part, stream, !ln_side,
                                             piggy ? (uint32_t*)(sv + L.st_sync) : nullptr, piggy ? d->F * 16 : 0));      // (+ clears the counters st_fused_bwd polls)
        if (ln_side) { TO_SIDE(); MAED_PROPAGATE(maed_layernorm_affine_finish(ln_part, M, C, g->ln2_g, g->ln2_b, wst)); }
        // attention: x_mid = x_in + proj(mix(x_s, x_t))
        TO_SIDE();
        WGRAD(maed_gemm_tn_wgrad(dxmid_tw, C, sv + L.mix, C, M, C, C, g->w_proj, C, g->b_proj, dmm, wst));
        PROF(PROF_GEMM_DGRAD, maed_gemm_nt(dxmid_tw, C, p->wt_proj, C, M, C, C, dmm, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));  // dmix
        if (st_fused(*d)) {   // dlogits + d(means) = dlogits . W_ts + dx_s / dx_t: one launch, dmix / x_s / x_t read once; the ts_attn weight gradient follows on the side stream
            // (the frames' arrival counters were cleared by the LayerNorm-2 backward kernel above)
            PROF(PROF_ST_BWD, maed_st_fused_bwd_ws(sc + S.act, sv + L.xs, sv + L.xt, logits, p->wt_ts, sc + S.dlog, sc + S.dxs, sc + S.dxt, (uint32_t*)(sv + L.st_sync),
                                                   (float*)(sv + L.st_ex), d->F, d->P, C, dt, !piggy, 0u, stream));
            TO_SIDE();
            WGRAD(maed_gemm_tn_wgrad(sc + S.dlog, 2 * C, sv + L.means, 2 * C, d->F, 2 * C, 2 * C, g->w_ts, 2 * C, g->b_ts, dmm, wst));
        } else {
            MAED_PROPAGATE(maed_st_mix_bwd_reduce(sc + S.act, sv + L.xs, sv + L.xt, logits, sc + S.dlog, (float*)(sc + S.ws), d->F, d->P, C, dt, stream));
            TO_SIDE();
            WGRAD(maed_gemm_tn_wgrad(sc + S.dlog, 2 * C, sv + L.means, 2 * C, d->F, 2 * C, 2 * C, g->w_ts, 2 * C, g->b_ts, dmm, wst));
            MAED_PROPAGATE(maed_gemm_nt(sc + S.dlog, 2 * C, p->wt_ts, 2 * C, d->F, 2 * C, 2 * C, dmm, MAED_EPI_STORE, nullptr, sc + S.dmeans, 2 * C, nullptr, nullptr, 0, 1, gi, stream));
            MAED_PROPAGATE(maed_st_mix_bwd_apply(sc + S.act, logits, sc + S.dmeans, sc + S.dxs, sc + S.dxt, d->F, d->P, C, dt, stream));
        }
        if (ss) MAED_HIP(hipStreamWaitEvent(main_s, ss->ev[63], 0), "ste_block_bwd: stream wait");               // the attention backward overwrites bigA, which the fc1 weight gradient reads
        PROF(PROF_ATTN_TM_BWD, maed_attn_temporal_bwd(sv + L.qkv, sv + L.xt, sc + S.dxt, (const float*)(sv + L.lse_t), sc + S.bigA, 0, d->F, d->P, d->H, d->T, scale, dt, stream));
        PROF(PROF_ATTN_SP_BWD, maed_attn_spatial_bwd(sv + L.qkv, sv + L.xs, sc + S.dxs, (const float*)(sv + L.lse_s), sc + S.bigA, 1, d->F, d->P, d->H, scale, dt,
                                                     MAED_IMPL_AUTO, stream));
        TO_SIDE();
        WGRAD(maed_gemm_tn_wgrad(sc + S.bigA, 3 * C, sv + L.ln1, C, M, 3 * C, C, g->w_qkv, C, g->b_qkv, dmm, wst));
        PROF(PROF_GEMM_DGRAD, maed_gemm_nt(sc + S.bigA, 3 * C, p->wt_qkv, 3 * C, M, C, 3 * C, dmm, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));
        MAED_PROPAGATE(maed_layernorm_bwd_ws(sc + S.act, dt, x_in, C, p->ln1_g, (const float*)(sv + L.mean1), (const float*)(sv + L.rstd1), dxmid, dx_in,
                                             dx_in_twin, g->ln1_g, g->ln1_b, M, C, ln_part1, stream, !ln_side));
        if (ln_side) { TO_SIDE(); MAED_PROPAGATE(maed_layernorm_affine_finish(ln_part1, M, C, g->ln1_g, g->ln1_b, wst)); }
        FROM_SIDE();                                                     // scratch and gradients: everything after this call sees the weight gradients
#undef TO_SIDE
#undef FROM_SIDE
#undef WGRAD
        return MAED_OK;
    }
    // ---- f32 parity mode (and impl = VALU): NT GEMMs on transposed copies ---------------------------------------
    // ---- MLP: x_out = x_mid + fc2(gelu(fc1(ln2(x_mid)))) ------------------------------------------------
    MAED_PROPAGATE(maed_transpose_cast(dx_out, MAED_F32, C, M, C, sc + S.dyt, Mp, sc + S.dyc, C, g->b_fc2, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sv + L.hact, dt, Hd, M, Hd, sc + S.bigT, Mp, nullptr, 0, nullptr, dt, stream));
    PROF(PROF_GEMM_WGRAD, wgrad(sc + S.dyt, sc + S.bigT, C, Hd, Mp, g->w_fc2, *d, stream));
    MAED_PROPAGATE(maed_gemm_nt(sc + S.dyc, C, p->wt_fc2, C, M, Hd, C, dt, MAED_EPI_MUL_DGELU, nullptr, sc + S.bigA, Hd, nullptr, sv + L.hpre, Hd, 1, gi, stream));
    MAED_PROPAGATE(maed_transpose_cast(sc + S.bigA, dt, Hd, M, Hd, sc + S.bigT, Mp, nullptr, 0, g->b_fc1, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sv + L.ln2, dt, C, M, C, sc + S.xT, Mp, nullptr, 0, nullptr, dt, stream));
    PROF(PROF_GEMM_WGRAD, wgrad(sc + S.bigT, sc + S.xT, Hd, C, Mp, g->w_fc1, *d, stream));
    MAED_PROPAGATE(maed_gemm_nt(sc + S.bigA, Hd, p->wt_fc1, Hd, M, C, Hd, dt, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));
    MAED_PROPAGATE(maed_layernorm_bwd(sc + S.act, dt, (const float*)(sv + L.xmid), C, p->ln2_g, (const float*)(sv + L.mean2), (const float*)(sv + L.rstd2),
                                      dx_out, dxmid, nullptr, g->ln2_g, g->ln2_b, M, C, stream));
    // ---- attention: x_mid = x_in + proj(mix(x_s, x_t)) ---------------------------------------------------
    MAED_PROPAGATE(maed_transpose_cast(dxmid, MAED_F32, C, M, C, sc + S.dyt, Mp, sc + S.dyc, C, g->b_proj, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sv + L.mix, dt, C, M, C, sc + S.xT, Mp, nullptr, 0, nullptr, dt, stream));
    PROF(PROF_GEMM_WGRAD, wgrad(sc + S.dyt, sc + S.xT, C, C, Mp, g->w_proj, *d, stream));
    MAED_PROPAGATE(maed_gemm_nt(sc + S.dyc, C, p->wt_proj, C, M, C, C, dt, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));  // dmix
    MAED_PROPAGATE(maed_st_mix_bwd_reduce(sc + S.act, sv + L.xs, sv + L.xt, logits, sc + S.dlog, (float*)(sc + S.ws), d->F, d->P, C, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sc + S.dlog, dt, 2 * C, d->F, 2 * C, sc + S.dlogT, Fp, nullptr, 0, g->b_ts, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sv + L.means, dt, 2 * C, d->F, 2 * C, sc + S.meansT, Fp, nullptr, 0, nullptr, dt, stream));
    PROF(PROF_GEMM_WGRAD, wgrad(sc + S.dlogT, sc + S.meansT, 2 * C, 2 * C, Fp, g->w_ts, *d, stream));
    MAED_PROPAGATE(maed_gemm_nt(sc + S.dlog, 2 * C, p->wt_ts, 2 * C, d->F, 2 * C, 2 * C, dt, MAED_EPI_STORE, nullptr, sc + S.dmeans, 2 * C, nullptr, nullptr, 0, 1, gi, stream));
    MAED_PROPAGATE(maed_st_mix_bwd_apply(sc + S.act, logits, sc + S.dmeans, sc + S.dxs, sc + S.dxt, d->F, d->P, C, dt, stream));
    PROF(PROF_ATTN_TM_BWD, maed_attn_temporal_bwd(sv + L.qkv, sv + L.xt, sc + S.dxt, (const float*)(sv + L.lse_t), sc + S.bigA, 0, d->F, d->P, d->H, d->T, scale, dt, stream));
    PROF(PROF_ATTN_SP_BWD, maed_attn_spatial_bwd(sv + L.qkv, sv + L.xs, sc + S.dxs, (const float*)(sv + L.lse_s), sc + S.bigA, 1, d->F, d->P, d->H, scale, dt,
                                         d->impl == MAED_IMPL_VALU ? MAED_IMPL_VALU : MAED_IMPL_AUTO, stream));
    MAED_PROPAGATE(maed_transpose_cast(sc + S.bigA, dt, 3 * C, M, 3 * C, sc + S.bigT, Mp, nullptr, 0, g->b_qkv, dt, stream));
    MAED_PROPAGATE(maed_transpose_cast(sv + L.ln1, dt, C, M, C, sc + S.xT, Mp, nullptr, 0, nullptr, dt, stream));
    PROF(PROF_GEMM_WGRAD, wgrad(sc + S.bigT, sc + S.xT, 3 * C, C, Mp, g->w_qkv, *d, stream));
    MAED_PROPAGATE(maed_gemm_nt(sc + S.bigA, 3 * C, p->wt_qkv, 3 * C, M, C, 3 * C, dt, MAED_EPI_STORE, nullptr, sc + S.act, C, nullptr, nullptr, 0, 1, gi, stream));
    MAED_PROPAGATE(maed_layernorm_bwd(sc + S.act, dt, x_in, C, p->ln1_g, (const float*)(sv + L.mean1), (const float*)(sv + L.rstd1), dxmid, dx_in, dx_in_twin,
                                      g->ln1_g, g->ln1_b, M, C, stream));
    return MAED_OK;
}

// ---- error plumbing / version ------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void maed_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* maed_last_error(void) { return g_err; }
#ifdef MAED_HOSTSIM
extern "C" int maed_version(void) { return -100; }   // x86 build for tests/hostsim: never the product library
#else
extern "C" int maed_version(void) { return 100; }
#endif
