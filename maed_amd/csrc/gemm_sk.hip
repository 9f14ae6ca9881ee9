// Persistent K-stream GEMM (round 6): the nn.Linear family  out = epilogue(A[M,K] . B[N,K]^T)  (bf16, fp32 accumulation) as ONE workgroup
// per CU that walks a list of work items (output tile, K range) as one continuous stream of K tiles.
// Replaces, for the large-M shapes of the STE, the per-tile launches of gemm.hip (128 x 128) / gemm256.hip (256 x 256):
// vision_transformer.py:98-111 (Mlp fc1 / fc2), :124-128,147 (qkv), :176 (proj) and their input gradients through autograd.
//
// Why (rounds 3-5; profiles/r06_vendor_kernels.txt): at 128 x 128 tiles these GEMMs are bound by L2 -> LDS operand
// traffic; the 256 x 256 pipeline of gemm256.hip halves that traffic but pays a ~2 us prologue and an un-overlapped epilogue per tile
// (a K = 512 tile is 8 K tiles = ~10 us of main loop), and its grids quantise badly (qkv: 594 tiles = 2.32 waves of 256 CUs; fc2: 198
// tiles = 0.77).  The vendor library's winners at these shapes are stream-K kernels on 256 x 256 x 64 macro tiles.  Here:
//   * the main loop IS gemm256.hip's (same LDS images, same four-phase K tile, same counted-vmcnt LDS-DMA ring, waves 4-7 one barrier
//     behind waves 0-3), but the DMA issue side runs through item boundaries: the first seven half-tiles of the NEXT item are in flight
//     or landed when the current item's last MFMA retires -- no prologue bubble between tiles;
//   * the epilogue of an item runs between two K tiles of that stream, through LDS the ring does not need at that moment (slot A1 of
//     buffer 1, whose next DMA is issued in the first phase after the epilogue, for waves 0-3; 16 KB beside the ring for waves 4-7),
//     wave-private (no barrier), full 128-byte row segments -- the fused epilogues of gemm_epilogue.cuh unchanged;
//   * stream-K: tiles that do not fill a whole round of the grid are cut along K into contiguous ranges of K-tile PAIRS, one range per
//     workgroup.  A tile cut into parts is finished by the workgroup that holds its LAST part (highest workgroup id), which processes it
//     as its LAST item; every other part is the FIRST item of its workgroup, which stores its accumulators to a slab (256 KB, register
//     image: fully coalesced) and publishes a flag.  A finisher therefore only ever waits for lower-numbered, earlier-dispatched workgroups
//     whose slab was due long before (no mutual waiting: safe when another kernel holds part of the chip), sums the slabs in a fixed
//     order (deterministic: no atomics) and runs the normal epilogue.  Hand-off: write-through (sc1) stores -> vmcnt(0) -> workgroup rendezvous ->
//     one lane: relaxed agent-scope flag store; finisher: one lane polls relaxed (bounded: a timeout poisons nothing silently,
//     it raises the library's fault word), agent-scope acquire, rendezvous, plain loads (MI355X_MICROARCH.md, inter-workgroup visibility).
// Items hold an even number of K tiles (K % 128 == 0, cuts at pair boundaries), so an item always starts in LDS buffer 0.
#include "common.cuh"
#include "gemm_epilogue.cuh"
#include <mutex>

#define SK_T 256
#define SK_BK 64
#define SK_SLOT (128 * SK_BK)          // elements per half-tile slot (16 KB)
#define SK_A0 0                        // slot order inside a buffer: consumption order (as gemm256.hip)
#define SK_B0 1
#define SK_B1 2
#define SK_A1 3
#define SK_RING_ELEMS (2 * 4 * SK_SLOT)             // 128 KB
#define SK_LDS_ELEMS (SK_RING_ELEMS + 4 * 2048)     // + 16 KB: epilogue staging of waves 4-7
#define SK_STAGE_BYTE0 (7 * SK_SLOT * 2)            // staging of wave w: byte offset SK_STAGE_BYTE0 + w * 4096 (buffer 1 slot A1, then the extra 16 KB)
#define SK_SLAB_FLOATS (SK_T * SK_T)                // one partial tile (fp32)
#define SK_FULL 0
#define SK_WRITE 1
#define SK_FINISH 2

struct SkPlan {
    int tiles, tiles_n;        // output tiles (256 x 256), tiles per row of tiles
    int kp;                    // K-tile pairs per tile (K / 128)
    int dp_tiles;              // tiles [0, dp_tiles) are taken whole, tile g + i * grid by workgroup g; the rest is cut into K ranges
    uint32_t epoch;            // flag value of this launch
};

#ifdef MAED_HOSTSIM
static inline uint32_t sk_flag_load(const uint32_t* p) { return *p; }
static inline void sk_flag_store(uint32_t* p, uint32_t v) { *p = v; }
static inline void sk_slab_store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
static inline void sk_acquire() {}
static inline void sk_sleep() {}
#define SK_SPIN_LIMIT 1u
#else
__device__ __forceinline__ uint32_t sk_flag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sk_flag_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// slab store: 16 bytes per lane, write-through (sc0 sc1): the data leaves the XCD's L2 with the store, so publishing needs no L2 write-back -- an agent-scope release
// (buffer_wbl2) would write back EVERY dirty line of the XCD's L2, other workgroups' output tiles included (MI355X_MICROARCH.md: publish-large 3.0 vs 8.2 us)
__device__ __forceinline__ void sk_slab_store(float* p, float4 v) {
    typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
    const sk_f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void sk_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void sk_sleep() { __builtin_amdgcn_s_sleep(16); }
#define SK_SPIN_LIMIT (1u << 22)
#endif

// what workgroup g does, in processing order (all values wave-uniform)
struct SkWork {
    int g, grid, n_dp, n_sk, n_items, has_w;
    uint32_t sk_a, sk_b;       // this workgroup's range of K-tile pairs inside the stream-K region
    int tl_hi, tl_lo;          // first / last stream-K tile it touches (relative to dp_tiles)
};
struct SkItem { int tile, p0, np, kind; };
// wave-uniform values the compiler must keep in SGPRs (loop-carried values of the item bookkeeping end up in VGPRs otherwise)
#define SK_UNI(x_) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x_)))
#ifdef MAED_HOSTSIM
#define SK_OPAQUE(v_) ((void)0)
#else
#define SK_OPAQUE(v_) asm volatile("" : "+v"(v_))
#endif

__device__ __forceinline__ void sk_range(const SkPlan& P, int grid, int g, uint32_t& a, uint32_t& b) {
    const uint32_t skp = (uint32_t)(P.tiles - P.dp_tiles) * (uint32_t)P.kp;       // (grid + 1) * skp < 2^32 (launcher)
    a = ((uint32_t)g * skp) / (uint32_t)grid;
    b = ((uint32_t)(g + 1) * skp) / (uint32_t)grid;
}
__device__ __forceinline__ SkWork sk_work(const SkPlan& P, int grid, int g) {
    SkWork w;
    w.g = g; w.grid = grid;
    w.n_dp = g < P.dp_tiles ? (P.dp_tiles - g + grid - 1) / grid : 0;
    sk_range(P, grid, g, w.sk_a, w.sk_b);
    const bool has = w.sk_b > w.sk_a;
    w.tl_hi = has ? (int)((w.sk_b - 1) / (uint32_t)P.kp) : 0;
    w.tl_lo = has ? (int)(w.sk_a / (uint32_t)P.kp) : 0;
    w.n_sk = has ? w.tl_hi - w.tl_lo + 1 : 0;
    w.has_w = has && (w.sk_b != (uint32_t)(w.tl_hi + 1) * (uint32_t)P.kp);      // its highest tile does not end inside this range: somebody else finishes it
    w.n_items = w.n_dp + w.n_sk;
    w.n_dp = (int)SK_UNI(w.n_dp); w.n_sk = (int)SK_UNI(w.n_sk); w.n_items = (int)SK_UNI(w.n_items); w.has_w = (int)SK_UNI(w.has_w);
    w.sk_a = SK_UNI(w.sk_a); w.sk_b = SK_UNI(w.sk_b); w.tl_hi = (int)SK_UNI(w.tl_hi); w.tl_lo = (int)SK_UNI(w.tl_lo);
    return w;
}
// item j of the processing order: [the part somebody else finishes] [whole tiles of the data-parallel region] [stream-K tiles, descending: the last one may be
// a tile this workgroup finishes]
__device__ __forceinline__ SkItem sk_item(const SkPlan& P, const SkWork& w, int j) {
    SkItem it;
    int tl;
    if (w.has_w && j == 0) tl = w.tl_hi;
    else {
        const int jj = j - w.has_w;
        if (jj < w.n_dp) { it.tile = (int)SK_UNI(w.g + jj * w.grid); it.p0 = 0; it.np = P.kp; it.kind = SK_FULL; return it; }
        tl = w.tl_hi - w.has_w - (jj - w.n_dp);
    }
    const uint32_t t0 = (uint32_t)tl * (uint32_t)P.kp, t1 = t0 + (uint32_t)P.kp;
    const uint32_t a = w.sk_a > t0 ? w.sk_a : t0, b = w.sk_b < t1 ? w.sk_b : t1;
    it.tile = P.dp_tiles + tl; it.p0 = (int)(a - t0); it.np = (int)(b - a);
    it.kind = (b != t1) ? SK_WRITE : (a != t0) ? SK_FINISH : SK_FULL;
    it.tile = (int)SK_UNI(it.tile); it.p0 = (int)SK_UNI(it.p0); it.np = (int)SK_UNI(it.np); it.kind = (int)SK_UNI(it.kind);
    return it;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_sk_bf16_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                                  int64_t M, int64_t N, int64_t K, SkPlan P, EpiArgs e,
                                                                  float* __restrict__ slabs, uint32_t* __restrict__ flags, uint32_t* fault) {
    __shared__ __attribute__((aligned(1024))) unsigned short lds_raw[SK_LDS_ELEMS];          // 144 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: LDS-DMA bases (M0) and the wave-group branches stay on the SALU
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int grid = (int)gridDim.x;
#ifdef MAED_HOSTSIM
    const int g = (int)blockIdx.x;                  // (the simulator runs workgroups in blockIdx order: a finisher's predecessors must have run)
#else
    const int g = xcd_remap((int)blockIdx.x, grid); // consecutive ids share an XCD: the tiles of one A panel / the parts of one tile meet in one L2
#endif
    const SkWork W = sk_work(P, grid, g);
    if (W.n_items == 0) return;

    // ---- staging map (gemm256.hip): a half-tile is 128 rows x 128 B = 1024 chunks of 16 B, two per thread (round i = 0, 1); wave w fills slot rows
    //      8w + 64i .. +7, lane l the (swizzled) chunk of row 8w + 64i + (l>>3).  Slot row s of A-half q is tile row (s>>6)*128 + q*64 + (s&63);
    //      slot row s of B-half q is tile column (s>>5)*64 + q*32 + (s&31).
    const int r = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((r >> 1) & 7);
    uint32_t ao0, ao1, ao2, ao3, bo0, bo1, bo2, bo3;                  // index 2*i + q; BYTE offsets of the ISSUE side's item (host-checked < 4 GB)
    uint32_t pa1, pa3;                                                // A1 offsets of the K tile whose A1 half is still to be issued
    // (32-bit arithmetic: rows < 2^31, byte offsets < 4 GB -- launcher; r_ / sc_ are opaque copies of the lane's row / chunk so that the eight row terms are
    //  recomputed per item instead of living in sixteen registers across the K stream)
    const uint32_t lda2 = (uint32_t)lda * 2u, ldb2 = (uint32_t)ldb * 2u;
    const int Mm1 = (int)M - 1, Nm1 = (int)N - 1;
#define SK_OFFS(j, m0_, n0_)                                                                        \
    {                                                                                               \
        const int i_ = (j) >> 1, q_ = (j) & 1;                                                      \
        int ar = (m0_) + i_ * 128 + q_ * 64 + r_;                                                   \
        int br = (n0_) + ((r_ >> 5) + 2 * i_) * 64 + q_ * 32 + (r_ & 31);                           \
        ar = ar > Mm1 ? Mm1 : ar;                                                                   \
        br = br > Nm1 ? Nm1 : br;                                                                   \
        ao##j = (uint32_t)ar * lda2 + sc_; bo##j = (uint32_t)br * ldb2 + sc_;                       \
    }
#define SK_SET_OFFS(tile_)                                                                          \
    {                                                                                               \
        const int m0__ = ((tile_) / P.tiles_n) * SK_T, n0__ = ((tile_) % P.tiles_n) * SK_T;         \
        int r_ = r; uint32_t sc_ = (uint32_t)schunk * 16u;                                          \
        SK_OPAQUE(r_); SK_OPAQUE(sc_);                                                              \
        SK_OFFS(0, m0__, n0__) SK_OFFS(1, m0__, n0__) SK_OFFS(2, m0__, n0__) SK_OFFS(3, m0__, n0__)   \
    }
    unsigned short* const ldsw = lds_raw + wave * 8 * SK_BK;          // this wave's rows of round 0 inside a slot (scalar)
    const char* const Ab = reinterpret_cast<const char*>(A);
    const char* const Bb = reinterpret_cast<const char*>(B);
    // one LDS-DMA per thread: scalar base (operand + K byte offset) + 32-bit lane offset -> round i_ of slot slot_ of buffer buf_
#define SK_DMA(base_, kb_, off_, buf_, slot_, i_) MAED_LDS_DMA16((base_) + SK_UNI(kb_), off_, ldsw + ((buf_) * 4 + (slot_)) * SK_SLOT + (i_) * 64 * SK_BK)

    // ---- fragments (gemm256.hip): A rows wr*64 + rt*32 + l31 of slot A[qm], B rows wc*32 + l31 of slot B[qn]; chunk (2*kk + hi) ^ fsw
    const int fsw = (l31 >> 1) & 7;
    const char* const ldsb = reinterpret_cast<const char*>(lds_raw);
    const char* const fa0 = ldsb + (wr * 64 + l31) * (SK_BK * 2) + ((0 + hi) ^ fsw) * 16;
    const char* const fa1 = ldsb + (wr * 64 + l31) * (SK_BK * 2) + ((2 + hi) ^ fsw) * 16;
    const char* const fa2 = ldsb + (wr * 64 + l31) * (SK_BK * 2) + ((4 + hi) ^ fsw) * 16;
    const char* const fa3 = ldsb + (wr * 64 + l31) * (SK_BK * 2) + ((6 + hi) ^ fsw) * 16;
    const char* const fb0 = ldsb + (wc * 32 + l31) * (SK_BK * 2) + ((0 + hi) ^ fsw) * 16;
    const char* const fb1 = ldsb + (wc * 32 + l31) * (SK_BK * 2) + ((2 + hi) ^ fsw) * 16;
    const char* const fb2 = ldsb + (wc * 32 + l31) * (SK_BK * 2) + ((4 + hi) ^ fsw) * 16;
    const char* const fb3 = ldsb + (wc * 32 + l31) * (SK_BK * 2) + ((6 + hi) ^ fsw) * 16;
    bf16x8_t a00, a01, a02, a03, a10, a11, a12, a13;            // a[rt][kk]   (named scalars: never demoted to scratch)
    bf16x8_t b00, b01, b02, b03, b10, b11, b12, b13;            // b[qn][kk]
    f32x16_t c000, c001, c010, c011, c100, c101, c110, c111;    // c[qm][rt][qn]
#define SK_ZERO_ACC()                                                                               \
    _Pragma("unroll") for (int x = 0; x < 16; ++x) { c000[x] = 0.f; c001[x] = 0.f; c010[x] = 0.f; c011[x] = 0.f; c100[x] = 0.f; c101[x] = 0.f; c110[x] = 0.f; c111[x] = 0.f; }
    SK_ZERO_ACC()
#define SK_FRAG(base_, buf_, slot_, rowoff_) (*reinterpret_cast<const bf16x8_t*>((base_) + (((buf_) * 4 + (slot_)) * SK_SLOT + (rowoff_) * SK_BK) * 2))
#define SK_READ_A(buf_, slot_)                                                                                              \
    a00 = SK_FRAG(fa0, buf_, slot_, 0); a01 = SK_FRAG(fa1, buf_, slot_, 0); a02 = SK_FRAG(fa2, buf_, slot_, 0); a03 = SK_FRAG(fa3, buf_, slot_, 0); \
    a10 = SK_FRAG(fa0, buf_, slot_, 32); a11 = SK_FRAG(fa1, buf_, slot_, 32); a12 = SK_FRAG(fa2, buf_, slot_, 32); a13 = SK_FRAG(fa3, buf_, slot_, 32);
#define SK_READ_B0(buf_) b00 = SK_FRAG(fb0, buf_, SK_B0, 0); b01 = SK_FRAG(fb1, buf_, SK_B0, 0); b02 = SK_FRAG(fb2, buf_, SK_B0, 0); b03 = SK_FRAG(fb3, buf_, SK_B0, 0);
#define SK_READ_B1(buf_) b10 = SK_FRAG(fb0, buf_, SK_B1, 0); b11 = SK_FRAG(fb1, buf_, SK_B1, 0); b12 = SK_FRAG(fb2, buf_, SK_B1, 0); b13 = SK_FRAG(fb3, buf_, SK_B1, 0);
    // one phase (gemm256.hip): fragment reads ; barrier ; fragments landed ; 8 MFMAs (transposed tiles: first operand = weight rows, a lane owns one output ROW)
    // at raised priority with the phase's two LDS-DMA instructions in their shadow ; counted wait ; barrier
#define SK_PHASE(READS_, ISSUE0_, ISSUE1_, WAIT_, c0_, c1_, bq_)                                     \
    READS_                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                   \
    MAED_WAIT_LGKMCNT0();                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a00, c0_, 0, 0, 0);                       \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##0, a10, c1_, 0, 0, 0);                       \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    ISSUE0_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a01, c0_, 0, 0, 0);                       \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##1, a11, c1_, 0, 0, 0);                       \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a02, c0_, 0, 0, 0);                       \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    ISSUE1_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##2, a12, c1_, 0, 0, 0);                       \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a03, c0_, 0, 0, 0);                       \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq_##3, a13, c1_, 0, 0, 0);                       \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    WAIT_;                                                                                          \
    __builtin_amdgcn_s_barrier();
#define SK_NONE ((void)0)
    // K-tile pair (t, t+1) of the stream, t in buffer 0.  Issue order = consumption order, six to seven phases ahead of the read (gemm256.hip):
    //   even tile  q1: A1(t+1) [pending: pa1 / pa3 at kpa]   q2..q4: A0, B0, B1 of tile t+2 [issue context at kb_nx]
    //   odd tile   q1: A1(t+2)                               q2..q4: A0, B0, B1 of tile t+3 [kb_nx + one K tile]
    // Every phase ends with vmcnt(8): the half-tile issued four phases ago has landed; it is read two phases later at the earliest.
    // EVENWAIT_: the wait of the even tile's four phases -- nothing for the FIRST pair of an item: whatever was issued before the item boundary has landed
    // (the prologue, the epilogue and the slab hand-over begin with vmcnt(0)), so the stream runs five phases (~2 us) before the first counted wait, which, VMEM
    // operations retiring in order, is also the first that has to see the epilogue's global stores acknowledged
#define SK_PAIR(EVENWAIT_)                                                                                                                                 \
    {                                                                                                                                                     \
        const uint32_t kb1__ = SK_UNI(kb_nx + SK_BK * 2);                                                                                                 \
        SK_PHASE(SK_READ_A(0, SK_A0) SK_READ_B0(0), SK_DMA(Ab, kpa, pa1, 1, SK_A1, 0), SK_DMA(Ab, kpa, pa3, 1, SK_A1, 1), EVENWAIT_, c000, c010, b0) \
        SK_PHASE(SK_READ_B1(0), SK_DMA(Ab, kb_nx, ao0, 0, SK_A0, 0), SK_DMA(Ab, kb_nx, ao2, 0, SK_A0, 1), EVENWAIT_, c001, c011, b1)              \
        SK_PHASE(SK_READ_A(0, SK_A1), SK_DMA(Bb, kb_nx, bo0, 0, SK_B0, 0), SK_DMA(Bb, kb_nx, bo2, 0, SK_B0, 1), EVENWAIT_, c101, c111, b1)        \
        SK_PHASE(, SK_DMA(Bb, kb_nx, bo1, 0, SK_B1, 0), SK_DMA(Bb, kb_nx, bo3, 0, SK_B1, 1), EVENWAIT_, c100, c110, b0)                           \
        SK_PHASE(SK_READ_A(1, SK_A0) SK_READ_B0(1), SK_DMA(Ab, kb_nx, ao1, 0, SK_A1, 0), SK_DMA(Ab, kb_nx, ao3, 0, SK_A1, 1), MAED_WAIT_VMCNT(8), c000, c010, b0) \
        SK_PHASE(SK_READ_B1(1), SK_DMA(Ab, kb1__, ao0, 1, SK_A0, 0), SK_DMA(Ab, kb1__, ao2, 1, SK_A0, 1), MAED_WAIT_VMCNT(8), c001, c011, b1)              \
        SK_PHASE(SK_READ_A(1, SK_A1), SK_DMA(Bb, kb1__, bo0, 1, SK_B0, 0), SK_DMA(Bb, kb1__, bo2, 1, SK_B0, 1), MAED_WAIT_VMCNT(8), c101, c111, b1)        \
        SK_PHASE(, SK_DMA(Bb, kb1__, bo1, 1, SK_B1, 0), SK_DMA(Bb, kb1__, bo3, 1, SK_B1, 1), MAED_WAIT_VMCNT(8), c100, c110, b0)                           \
        pa1 = ao1; pa3 = ao3; kpa = kb1__;                                                                                                                \
    }
    // ---- the issue side: (item, pair) of the next K-tile pair to copy
    int ij = 0;                                        // item the issue side is in
    SkItem iti = sk_item(P, W, 0);
    int left_i = iti.np;                               // pairs of that item not yet issued
    uint32_t kb_nx = SK_UNI((uint32_t)iti.p0 * (2 * SK_BK * 2));   // K byte offset of the pair to issue next (kept scalar: the DMA's base is an SGPR pair)
    uint32_t kpa;
    SK_SET_OFFS(iti.tile)
    // advance by one pair: inside the item, or into the next item (new operand offsets).  Past the end of this workgroup's stream the issue side stays where it
    // is: the last pair(s) of the stream copy the stream's last pair once more (112 KB per workgroup and launch, L2 hits, into ring slots nobody reads again) --
    // which keeps EVERY pair of the stream on the one steady-state code path: no drain variant, no second path that writes the accumulators
#define SK_ADVANCE()                                                                                \
    {                                                                                               \
        if (left_i > 1) { --left_i; kb_nx = SK_UNI(kb_nx + 2 * SK_BK * 2); }                        \
        else if (ij + 1 < W.n_items) {                                                              \
            ++ij;                                                                                   \
            iti = sk_item(P, W, ij);                                                                \
            left_i = iti.np; kb_nx = SK_UNI((uint32_t)iti.p0 * (2 * SK_BK * 2));                    \
            SK_SET_OFFS(iti.tile)                                                                   \
        }                                                                                           \
    }
    // ---- prologue: the first pair except the A1 half of its odd tile (the issue order of the steady state); A0, B0, B1 of its even tile must have landed
    {
        const uint32_t kb1 = SK_UNI(kb_nx + SK_BK * 2);
        SK_DMA(Ab, kb_nx, ao0, 0, SK_A0, 0); SK_DMA(Ab, kb_nx, ao2, 0, SK_A0, 1);
        SK_DMA(Bb, kb_nx, bo0, 0, SK_B0, 0); SK_DMA(Bb, kb_nx, bo2, 0, SK_B0, 1);
        SK_DMA(Bb, kb_nx, bo1, 0, SK_B1, 0); SK_DMA(Bb, kb_nx, bo3, 0, SK_B1, 1);
        SK_DMA(Ab, kb_nx, ao1, 0, SK_A1, 0); SK_DMA(Ab, kb_nx, ao3, 0, SK_A1, 1);
        SK_DMA(Ab, kb1, ao0, 1, SK_A0, 0); SK_DMA(Ab, kb1, ao2, 1, SK_A0, 1);
        SK_DMA(Bb, kb1, bo0, 1, SK_B0, 0); SK_DMA(Bb, kb1, bo2, 1, SK_B0, 1);
        SK_DMA(Bb, kb1, bo1, 1, SK_B1, 0); SK_DMA(Bb, kb1, bo3, 1, SK_B1, 1);
        pa1 = ao1; pa3 = ao3; kpa = kb1;
    }
    SK_ADVANCE()
    MAED_WAIT_VMCNT0();                               // (all seven half-tiles: the first pair of an item runs without waits)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind (wave-uniform branch)

    constexpr bool vec_ok = true;       // launcher: N, ldo, ldaux multiples of 8, 16-byte aligned outputs -- the scalar tails of gemm_epilogue.cuh are not instantiated here
    char* const stg = reinterpret_cast<char*>(lds_raw) + SK_STAGE_BYTE0 + wave * 4096;          // 16 rows x 256 B, 16-byte chunk c of row r at slot c ^ r
    const int r16 = l31 & 15, rhalf = l31 >> 4;
    const int rr = lane >> 3, c8 = lane & 7;
    float* const slab = slabs + (int64_t)g * SK_SLAB_FLOATS;

    for (int j = 0; j < W.n_items; ++j) {
        const SkItem it = sk_item(P, W, j);
        SK_PAIR(SK_NONE)
        SK_ADVANCE()
        for (int p = 1; p < it.np; ++p) {
            SK_PAIR(MAED_WAIT_VMCNT(8))
            SK_ADVANCE()
        }
        // (compiler fence: the accumulators below are final -- keeps the boundary code behind the last phase)
        __builtin_amdgcn_sched_barrier(0);
        // the lane indices of the boundary code, made opaque once per item: everything derived from them (32 row offsets times the leading dimension, 32 slab
        // addresses, the staging addresses) would otherwise be hoisted out of the item loop and live -- spilled -- across the K stream
        int rr_l = rr, c8_l = c8, tid_l = tid, r16_l = r16;
        SK_OPAQUE(rr_l); SK_OPAQUE(c8_l); SK_OPAQUE(tid_l); SK_OPAQUE(r16_l);

        // ---- the last part of a tile cut along K (always this workgroup's last item): add the slabs of the workgroups below that
        //      hold its other parts, nearest first.  (One loop for both kinds of item -- no slabs for a whole tile: accumulators that one branch modifies and
        //      the other does not would be copied, and spilled, at the merge.)
        int npart = 0;
        if (it.kind == SK_FINISH) {
            if (wr == 0) __builtin_amdgcn_s_barrier();     // re-align the two wave groups
            const uint32_t t0 = (uint32_t)(it.tile - P.dp_tiles) * (uint32_t)P.kp;
            for (int gp = g - 1; gp >= 0; --gp) {
                uint32_t pa_, pb_;
                sk_range(P, grid, gp, pa_, pb_);
                if (pb_ <= t0) break;
                if (pb_ > pa_) ++npart;
            }
            if (tid == 0) {
                bool ok = true;
                int seen = 0;
                for (int gp = g - 1; gp >= 0 && seen < npart; --gp) {
                    uint32_t pa_, pb_;
                    sk_range(P, grid, gp, pa_, pb_);
                    if (pb_ <= pa_) continue;
                    ++seen;
                    uint32_t spins = 0;
                    while (sk_flag_load(flags + gp) != P.epoch) {
                        sk_sleep();
                        if (++spins > SK_SPIN_LIMIT) { ok = false; break; }
                    }
                }
                if (!ok) maed_report_fault(fault);
                sk_acquire();
            }
            __syncthreads();
            npart = (int)SK_UNI(npart);
        }
        for (int gp = g - 1, seen = 0; seen < npart; --gp) {
            uint32_t pa_, pb_;
            sk_range(P, grid, gp, pa_, pb_);
            if (pb_ <= pa_) continue;
            ++seen;
            const float* const ps = slabs + (int64_t)gp * SK_SLAB_FLOATS;
#define SK_SLAB_LD4(a_, k_) const float4 v##a_##k_ = *reinterpret_cast<const float4*>(ps + ((int64_t)((a_) * 4 + (k_)) * 512 + tid_l) * 4);
#define SK_SLAB_AD4(acc_, a_, k_) acc_[4 * (k_)] += v##a_##k_.x; acc_[4 * (k_) + 1] += v##a_##k_.y; acc_[4 * (k_) + 2] += v##a_##k_.z; acc_[4 * (k_) + 3] += v##a_##k_.w;
#define SK_SLAB_LD(a_) SK_SLAB_LD4(a_, 0) SK_SLAB_LD4(a_, 1) SK_SLAB_LD4(a_, 2) SK_SLAB_LD4(a_, 3)
#define SK_SLAB_AD(acc_, a_) SK_SLAB_AD4(acc_, a_, 0) SK_SLAB_AD4(acc_, a_, 1) SK_SLAB_AD4(acc_, a_, 2) SK_SLAB_AD4(acc_, a_, 3)
            // sixteen 16-byte loads in flight per lane (the fragment registers are free here): two round trips per slab
            {
                __builtin_amdgcn_sched_barrier(0);
                SK_SLAB_LD(0) SK_SLAB_LD(1) SK_SLAB_LD(2) SK_SLAB_LD(3)
                SK_SLAB_AD(c000, 0) SK_SLAB_AD(c001, 1) SK_SLAB_AD(c010, 2) SK_SLAB_AD(c011, 3)
                __builtin_amdgcn_sched_barrier(0);
                SK_SLAB_LD(4) SK_SLAB_LD(5) SK_SLAB_LD(6) SK_SLAB_LD(7)
                SK_SLAB_AD(c100, 4) SK_SLAB_AD(c101, 5) SK_SLAB_AD(c110, 6) SK_SLAB_AD(c111, 7)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef SK_SLAB_LD4
#undef SK_SLAB_AD4
#undef SK_SLAB_AD
#undef SK_SLAB_LD
        }
        if (it.kind == SK_WRITE) {
            // ---- a part somebody else finishes: accumulators -> slab as a register image (store i of a lane at (i * 512 + tid) * 16 bytes: every wave store
            //      writes 1 KB contiguous), then publish.  Rendezvous of all eight waves: the groups are re-aligned, meet, and are staggered again.
#define SK_SLAB_ST(acc_, a_)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4)                                        \
                sk_slab_store(slab + ((int64_t)((a_) * 4 + q4) * 512 + tid_l) * 4, make_float4(acc_[4 * q4], acc_[4 * q4 + 1], acc_[4 * q4 + 2], acc_[4 * q4 + 3]));
            SK_SLAB_ST(c000, 0) SK_SLAB_ST(c001, 1) SK_SLAB_ST(c010, 2) SK_SLAB_ST(c011, 3) SK_SLAB_ST(c100, 4) SK_SLAB_ST(c101, 5) SK_SLAB_ST(c110, 6) SK_SLAB_ST(c111, 7)
#undef SK_SLAB_ST
            MAED_WAIT_VMCNT0();                        // this wave's slab stores (and its copies in flight) are done
            if (wr == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();              // all eight waves: the whole slab is in L2
            if (tid == 0) sk_flag_store(flags + g, P.epoch);          // (write-through slab stores, all drained: a relaxed agent-scope store publishes)
            if (wr == 1) __builtin_amdgcn_s_barrier();
        } else {
            // ---- epilogue: per (qm, rt) and row half a 16 x 64 fp32 piece of the wave's tile through its private staging area (lanes of the half write their
            //      eight float4, all lanes read full row segments back: 8 lanes cover 64 columns = one 128-byte line of bf16)
            const int64_t m0 = (int64_t)(it.tile / P.tiles_n) * SK_T, n0 = (int64_t)(it.tile % P.tiles_n) * SK_T;
            // Per (qm, rt): the operand the epilogue reads from memory (fp32 residual / bf16 pre-activation) for the piece's four row groups first -- one load latency
            // per 32 x 64 piece instead of one per row group, and they overlap the staging traffic -- then the two row halves through the staging area.
            constexpr bool kAux = EPI == MAED_EPI_RESID_F32 || EPI == MAED_EPI_MUL_DGELU;
#define SK_ROW(qm_, rt_, h_, ps_) (m0 + wr * 128 + (qm_) * 64 + (rt_) * 32 + (h_) * 16 + (ps_) * 8 + rr_l)
#define SK_AUX_LD(qm_, rt_, h_, ps_)                                                                                        \
            float ax##h_##ps_[8];                                                                                           \
            const bool hx##h_##ps_ = kAux && SK_ROW(qm_, rt_, h_, ps_) < M && c0 < N;                                        \
            {   /* unconditional loads at clamped coordinates: no control flow around the piece's sixteen registers */      \
                const int64_t rc__ = SK_ROW(qm_, rt_, h_, ps_) < M ? SK_ROW(qm_, rt_, h_, ps_) : M - 1, cc__ = c0 < N ? c0 : 0; \
                if constexpr (EPI == MAED_EPI_RESID_F32) ld8((const float*)e.aux + rc__ * e.ldaux + cc__, ax##h_##ps_);     \
                if constexpr (EPI == MAED_EPI_MUL_DGELU) ld8((const bf16*)e.aux + rc__ * e.ldaux + cc__, ax##h_##ps_);      \
            }
#define SK_APPLY(qm_, rt_, h_, ps_)                                                                                         \
            {                                                                                                               \
                const int lr = (ps_) * 8 + rr_l;                                                                            \
                const int64_t row = SK_ROW(qm_, rt_, h_, ps_);                                                              \
                const float4 u0 = *reinterpret_cast<const float4*>(stg + lr * 256 + (((2 * c8_l) ^ lr) << 4));              \
                const float4 u1 = *reinterpret_cast<const float4*>(stg + lr * 256 + (((2 * c8_l + 1) ^ lr) << 4));          \
                float v8[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};                                             \
                if (kAux && hx##h_##ps_) {                                                                                  \
                    if constexpr (EPI == MAED_EPI_RESID_F32) {                                                              \
                        if (e.bias) { float b8[8]; ld8(e.bias + c0, b8); _Pragma("unroll") for (int x = 0; x < 8; ++x) v8[x] += b8[x]; } \
                        _Pragma("unroll") for (int x = 0; x < 8; ++x) v8[x] += ax##h_##ps_[x];                              \
                        st8((float*)e.out + row * e.ldo + c0, v8);                                                          \
                    }                                                                                                       \
                    if constexpr (EPI == MAED_EPI_MUL_DGELU) {                                                              \
                        _Pragma("unroll") for (int x = 0; x < 8; ++x) v8[x] *= gelu_bwd<bf16>(ax##h_##ps_[x]);              \
                        st8((bf16*)e.out + row * e.ldo + c0, v8);                                                           \
                    }                                                                                                       \
                } else if (!kAux && row < M && c0 < N) epilogue_store8<EPI, bf16>(e, row, c0, N, v8, vec_ok);               \
            }
#define SK_STORE_HALF(accA_, accB_, qm_, rt_, h_)                                                                           \
            MAED_WAVE_LDS_SYNC();                                                                                           \
            if (rhalf == (h_)) {                                                                                            \
                _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                          \
                    *reinterpret_cast<float4*>(stg + r16_l * 256 + (((2 * q4 + hi) ^ r16_l) << 4)) = make_float4(accA_[4 * q4], accA_[4 * q4 + 1], accA_[4 * q4 + 2], accA_[4 * q4 + 3]);      \
                    *reinterpret_cast<float4*>(stg + r16_l * 256 + (((8 + 2 * q4 + hi) ^ r16_l) << 4)) = make_float4(accB_[4 * q4], accB_[4 * q4 + 1], accB_[4 * q4 + 2], accB_[4 * q4 + 3]);  \
                }                                                                                                           \
            }                                                                                                               \
            MAED_WAVE_LDS_SYNC();                                                                                           \
            SK_APPLY(qm_, rt_, h_, 0) SK_APPLY(qm_, rt_, h_, 1)
#define SK_STORE_PIECE(accA_, accB_, qm_, rt_)                                                                              \
            {                                                                                                               \
                __builtin_amdgcn_sched_barrier(0);     /* one scheduling region per piece: address arithmetic of later pieces hoisted up here costs registers */ \
                SK_AUX_LD(qm_, rt_, 0, 0) SK_AUX_LD(qm_, rt_, 0, 1) SK_AUX_LD(qm_, rt_, 1, 0) SK_AUX_LD(qm_, rt_, 1, 1)       \
                SK_STORE_HALF(accA_, accB_, qm_, rt_, 0) SK_STORE_HALF(accA_, accB_, qm_, rt_, 1)                           \
            }
            const int64_t c0 = n0 + wc * 64 + c8_l * 8;
            MAED_WAIT_VMCNT0();        // everything issued before this boundary has landed: the next item's first pair runs without waits
            SK_STORE_PIECE(c000, c001, 0, 0)
            SK_STORE_PIECE(c010, c011, 0, 1)
            SK_STORE_PIECE(c100, c101, 1, 0)
            SK_STORE_PIECE(c110, c111, 1, 1)
#undef SK_STORE_PIECE
#undef SK_STORE_HALF
#undef SK_APPLY
#undef SK_AUX_LD
#undef SK_ROW
            MAED_WAVE_LDS_SYNC();
        }
        SK_ZERO_ACC()
    }
    MAED_WAIT_VMCNT0();        // the re-issued copies of the stream's last pair must have landed before this workgroup's LDS is handed to the next one
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------
// Slabs + flags of the stream-K hand-off: ONE allocation made by maed_init_runtime (block.hip) -- SK_SLOTS independent sets, one per stream that launches this
// kernel (launches of one stream are serialised; two streams must not share slabs).  A stream beyond the table runs without K cuts.
#define SK_SLOTS 4
#define SK_MAX_GRID 512
namespace {
struct SkSlot { void* stream; uint32_t epoch; };
struct SkState {
    char* base = nullptr;                 // SK_SLOTS x (flags: SK_MAX_GRID x 4 B, padded to 4 KB ; slabs: ncu x 256 KB)
    int ncu = 0;
    int dev = -1;                         // the device the slabs live on (the one that was current at maed_init): launches on another device take the per-tile kernels
    int nslots = 0;
    SkSlot slot[SK_SLOTS];
    std::mutex mu;
};
SkState g_sk;
size_t sk_slot_bytes(int ncu) { return 4096 + (size_t)ncu * (SK_SLAB_FLOATS + 256) * sizeof(float); }      // (+ 256: the weight-gradient kernel's column sums, gemm_tn_sk.hip)
}  // namespace

// called once from maed_init_runtime (the one place that creates what the library owns); safe to call again
int maed_sk_init(void) {
    std::lock_guard<std::mutex> lk(g_sk.mu);
    if (g_sk.base) return MAED_OK;
#ifdef MAED_HOSTSIM
    g_sk.ncu = 8;
    g_sk.base = (char*)calloc(SK_SLOTS, sk_slot_bytes(g_sk.ncu));
    return g_sk.base ? MAED_OK : MAED_ERR_LAUNCH;
#else
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return MAED_ERR_LAUNCH; }
    g_sk.ncu = prop.multiProcessorCount;
    g_sk.dev = dev;
    if (g_sk.ncu < 8 || g_sk.ncu > SK_MAX_GRID) return MAED_ERR_UNSUPPORTED;
    void* p = nullptr;
    if (hipMalloc(&p, SK_SLOTS * sk_slot_bytes(g_sk.ncu)) != hipSuccess || !p) { (void)hipGetLastError(); return MAED_ERR_LAUNCH; }
    for (int s = 0; s < SK_SLOTS; ++s)
        if (hipMemset((char*)p + s * sk_slot_bytes(g_sk.ncu), 0, 4096) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return MAED_ERR_LAUNCH; }
    g_sk.base = (char*)p;
    return MAED_OK;
#endif
}
int maed_sk_cus(void) { return g_sk.ncu; }

// the slab set of a stream (assigned on first use) and the next epoch of that set; -1: table full, or the calling thread's device is not the slabs'
static int sk_take_slot(hipStream_t s, uint32_t* epoch) {
#ifndef MAED_HOSTSIM
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_sk.dev) { (void)hipGetLastError(); return -1; }
#endif
    std::lock_guard<std::mutex> lk(g_sk.mu);
    for (int i = 0; i < g_sk.nslots; ++i)
        if (g_sk.slot[i].stream == (void*)s) { *epoch = ++g_sk.slot[i].epoch; if (*epoch == 0) *epoch = ++g_sk.slot[i].epoch; return i; }
    if (g_sk.nslots == SK_SLOTS) return -1;
    const int i = g_sk.nslots++;
    g_sk.slot[i].stream = (void*)s; g_sk.slot[i].epoch = 1; *epoch = 1;
    return i;
}

// the slab set of `s` for the weight-gradient kernel (gemm_tn_sk.hip: one slab per workgroup, no flags); NULL: no allocation / more streams than sets
float* maed_sk_slab_set(hipStream_t s, size_t* bytes, int* ncu) {
    if (!g_sk.base && maed_sk_init() != MAED_OK) return nullptr;
    uint32_t epoch = 0;
    const int slot = sk_take_slot(s, &epoch);
    if (slot < 0) return nullptr;
    *bytes = sk_slot_bytes(g_sk.ncu) - 4096; *ncu = g_sk.ncu;
    return (float*)(g_sk.base + (size_t)slot * sk_slot_bytes(g_sk.ncu) + 4096);
}

template <int EPI>
static void launch_sk(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e, const SkPlan& P, int grid,
                      float* slabs, uint32_t* flags, hipStream_t s) {
    hipLaunchKernelGGL((gemm_nt_sk_bf16_kernel<EPI>), dim3((unsigned)grid), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)B, ldb, M, N, K, P, e, slabs, flags,
                       maed_fault_word());
}

// The decomposition.  T tiles on a grid of G workgroups: whole rounds of the grid are taken tile by tile (tile g + i G); what is left is cut along K when that
// pays -- every workgroup then carries one seam (a slab written or read: a few microseconds), so the K cuts are spread over at least one tile per workgroup
// ("two-tile" stream-K: the last whole round joins the remainder) and are skipped when the remainder nearly fills a round anyway.
//   mode 1 (default): as described;  2: never cut along K;  3: cut whenever the tiles do not fill whole rounds
// grid_opt: 0 = one workgroup per CU.
bool maed_gemm_nt_sk_shape_ok(int64_t M, int64_t N, int64_t K) { return K % 128 == 0 && K >= 128 && N >= 256 && N % 8 == 0 && M >= 256; }
static bool sk_epi_ok(const EpiArgs& e) {      // the kernel instantiates the 8-wide epilogues only
    return e.ldo % 8 == 0 && e.ldaux % 8 == 0 && is_aligned(e.out, 16) && is_aligned(e.out2, 16) && is_aligned(e.aux, 16) && is_aligned(e.bias, 16) && !e.twin && !e.lo;
}
bool maed_gemm_nt_sk_launch(int epilogue, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiArgs& e,
                            int mode, int grid_opt, hipStream_t s) {
    if (!sk_epi_ok(e)) return false;
    // mode 1, measured (profiles/r06_sk_micro.txt): the persistent stream wins where a tile is long -- K >= 2560: cfg5's fc2 / d(fc1), 174 vs 184 us -- and loses to the
    // per-tile kernels at the K = 512 ... 2048 shapes of cfg3 (one workgroup per CU cannot hide an item's epilogue stores behind another workgroup's MFMAs)
    if (mode == 1 && K < 2560) return false;
    if (!g_sk.base && maed_sk_init() != MAED_OK) return false;
    const int tm = (int)((M + SK_T - 1) / SK_T), tn = (int)((N + SK_T - 1) / SK_T);
    const int64_t T = (int64_t)tm * tn;
    if (T > (1 << 24)) return false;
    int G = grid_opt > 0 ? grid_opt : g_sk.ncu;
    if (G > g_sk.ncu) G = g_sk.ncu;
    SkPlan P;
    P.tiles = (int)T; P.tiles_n = tn; P.kp = (int)(K / 128); P.epoch = 0;
    const int rounds = (int)(T / G), rem = (int)(T % G);
    bool cut = mode != 2 && rem != 0;
    if (cut && mode == 1) cut = rem * 100 < G * 85;                 // a remainder that nearly fills a round: the seams cost more than the idle CUs
    if (cut && (uint64_t)(G + 1) * (uint64_t)(rem + G) * (uint64_t)P.kp >= (1ull << 32)) cut = false;   // (the kernel's 32-bit range arithmetic)
    float* slabs = nullptr;
    uint32_t* flags = nullptr;
    if (cut) {
        uint32_t epoch = 0;
        const int slot = sk_take_slot(s, &epoch);
        if (slot < 0) cut = false;
        else {
            char* b = g_sk.base + (size_t)slot * sk_slot_bytes(g_sk.ncu);
            flags = (uint32_t*)b; slabs = (float*)(b + 4096);
            P.epoch = epoch;
        }
    }
    if (cut) P.dp_tiles = rounds >= 1 ? (rounds - 1) * G : 0;      // the remainder + one whole round are cut; fewer tiles than workgroups: everything is
    else { P.dp_tiles = (int)T; if (T < G) G = (int)T; }
    switch (epilogue) {
        case MAED_EPI_STORE: launch_sk<MAED_EPI_STORE>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_GELU: launch_sk<MAED_EPI_GELU>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_RESID_F32: launch_sk<MAED_EPI_RESID_F32>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_MUL_DGELU: launch_sk<MAED_EPI_MUL_DGELU>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_STORE_F32: launch_sk<MAED_EPI_STORE_F32>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_TANH: launch_sk<MAED_EPI_TANH>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        case MAED_EPI_ADD: launch_sk<MAED_EPI_ADD>(A, lda, B, ldb, M, N, K, e, P, G, slabs, flags, s); return true;
        default: return false;
    }
}
