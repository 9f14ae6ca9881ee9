"""STE Block for the ablation st_modes 'series', 'vanilla', 'temporal' and 'coupling' (reference:
lib/models/vision_transformer.py:136-178, 244-261).

The configured mode ('parallel') runs as ONE fused library call per direction (ops.STEBlockFn); the other modes are not on
the benchmarked path and are composed here from the same libmaed_hip kernels, one autograd Function per stage:

    LayerNormFn        maed_layernorm_fwd / _bwd                fp32 residual stream -> compute-dtype rows
    LinearTokFn        maed_gemm_nt (+ maed_gemm_tn_wgrad)       nn.Linear on (rows, C); compute-dtype or fp32 output
    SpatialAttnFn      maed_attn_spatial_fwd / _bwd             reads q/k/v in place from the (F,P,3C) qkv rows
    TemporalAttnFn     maed_attn_temporal_fwd / _bwd
    TokenMeanFn        maed_st_colmean                          'temporal' mode's mean over the tokens of a frame
    MlpFn              fc1+GELU and fc2 with the fused epilogues (GELU', bias) of the fused block

'coupling' attends over the T*P tokens of a clip.  Frames of a clip are contiguous rows, and the reference's reshape_T
(:180-189) orders a clip's tokens (t, p), i.e. exactly the row-major order of the (T, P, 3C) qkv rows of that clip -- so
coupling is the spatial entry point on the VIEW (N, T*P, 3C), no data movement.  Past the whole-head kernels' limits (512
tokens forward, 320 backward; 16 x 197 = 3152 at cfg3) the library switches to the K/V-tiled long-sequence kernels of
csrc/attn_long.hip: MFMA flash forward / two-pass backward in bf16, exact VALU tiles in the f32 parity mode.
"""
import torch

from . import _lib as L
from . import ops


def _wgrad(dy, x, need_bias, prec=None):
    """(dW fp32 [N,K], db fp32 [N] or None) for y = x W^T + b; dy (M,N), x (M,K) in the same compute dtype; prec: the caller's own fp32 engine (ops.mm_code)"""
    N, K = dy.shape[1], x.shape[1]
    db = torch.zeros(N, dtype=torch.float32, device=dy.device) if need_bias else None
    if ops.lib_matmul_dtype(dy.dtype, prec) and N % 8 == 0 and K % 8 == 0:
        dW = torch.zeros(N, K, dtype=torch.float32, device=dy.device)
        ops.gemm_tn_wgrad(dy, x, dW, db, prec=prec)
        return dW, db
    dyt, _ = ops.transpose_cast(dy, dy.dtype, colsum=db)
    xt, _ = ops.transpose_cast(x, x.dtype)
    tiles = max(1, ((N + 127) // 128) * ((K + 127) // 128))
    dW = ops.gemm_nt(dyt, xt, L.EPI_ATOMIC_F32, splitk=max(1, min(dyt.shape[1] // 256, 1024 // tiles)))
    return dW, db


class _ApplyGradMode:
    """grad mode at apply() time (inside Function.forward it is always off; ctx.needs_input_grad ignores torch.no_grad())"""
    on = True


def _direct_begin(ctx, direct, idx, *params):
    """(params) if this forward's backward will run and may write their gradients into .grad itself, else None; counts the forward in"""
    if not direct or not _ApplyGradMode.on or not all(ctx.needs_input_grad[i] for i in idx if params[idx.index(i)] is not None):
        return None
    ops.direct_grad_begin(*params)
    return params


def _as(t, dtype):
    """contiguous copy-free view of t in `dtype` (casts only when needed)"""
    t = t if t.dtype == dtype else t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _wgrad_direct(dy, x, params, prec=None):
    """dW / db of y = x W^T + b accumulated straight into the parameters' .grad (True), or nothing done (False: the caller takes autograd's way)"""
    w, b = params
    N, K = dy.shape[1], x.shape[1]
    gw, gb = ops.direct_grad_slot(w), ops.direct_grad_slot(b)
    if gw is None or (b is not None and gb is None) or not (ops.lib_matmul_dtype(dy.dtype, prec) and N % 8 == 0 and K % 8 == 0):
        ops.direct_grad_cancel(*params)
        return False
    ops.gemm_tn_wgrad(dy, x, gw.view(N, K), gb, prec=prec)
    ops.direct_grad_done(*params)
    return True


class _ApplyMixin:
    @classmethod
    def apply(cls, *args):
        _ApplyGradMode.on = torch.is_grad_enabled()
        return super().apply(*args)


class LayerNormFn(_ApplyMixin, torch.autograd.Function):
    """fp32 rows (R,C) -> LayerNorm rows in `out_dtype` (vision_transformer.py:259-260 norm1/norm2)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype, direct=False):
        """direct: gamma / beta have this ONE use per forward -- their gradients may be accumulated straight into .grad (ops.direct_grad_slot)"""
        x = _as(x, torch.float32)
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, out_dtype, eps=eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.direct = _direct_begin(ctx, direct, (1, 2), gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        if ctx.direct is not None:
            gg, gb = ops.direct_grad_slot(ctx.direct[0]), ops.direct_grad_slot(ctx.direct[1])
            if gg is not None and gb is not None:
                dx, _, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=gg, dbeta=gb)
                ops.direct_grad_done(*ctx.direct)
                return dx.view_as(x), None, None, None, None, None
            ops.direct_grad_cancel(*ctx.direct)
        dx, dgamma, dbeta = ops.layernorm_bwd(dy, x, gamma, mean, rstd)
        return dx.view_as(x), dgamma, dbeta, None, None, None


class LinearTokFn(_ApplyMixin, torch.autograd.Function):
    """y = x W^T + b on (M,K) rows in the compute dtype; `out_f32` writes fp32 (the branch outputs that join the residual)"""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, out_f32, prec=None, direct=False):
        """prec: fp32 rows only -- the caller's own matrix-product engine ("bf16x3": the decoder head of a bf16-mode model, KTD.head_matmul);
        direct: weight / bias have this ONE use per forward -- their gradients may be accumulated straight into .grad (ops.direct_grad_slot)"""
        (wc, wt), = cache.get([weight], x.dtype)
        x = _as(x, x.dtype)
        ctx.save_for_backward(x)
        ctx.wt, ctx.has_bias, ctx.prec = wt, bias is not None, prec
        ctx.direct = _direct_begin(ctx, direct, (1, 2), weight, bias)
        return ops.gemm_nt(x, wc, L.EPI_STORE_F32 if out_f32 else L.EPI_STORE, bias=bias, prec=prec)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = _as(dy, x.dtype)
        dx = ops.gemm_nt(dy, ctx.wt, L.EPI_STORE, prec=ctx.prec) if ctx.needs_input_grad[0] else None
        if ctx.direct is not None and _wgrad_direct(dy, x, ctx.direct, ctx.prec):
            return dx, None, None, None, None, None, None
        dW, db = _wgrad(dy, x, ctx.has_bias, ctx.prec) if ctx.needs_input_grad[1] else (None, None)
        return dx, dW, db, None, None, None, None


class MlpFn(torch.autograd.Function):
    """fc2(GELU_erf(fc1(h))) (vision_transformer.py:96-112); h compute dtype (M,C) -> fp32 (M,C)"""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, cache):
        (w1c, w1t), (w2c, w2t) = cache.get([w1, w2], h.dtype)
        h = _as(h, h.dtype)
        act, pre = ops.gemm_nt(h, w1c, L.EPI_GELU, bias=b1)
        y = ops.gemm_nt(act, w2c, L.EPI_STORE_F32, bias=b2)
        ctx.save_for_backward(h, act, pre)
        ctx.wts = (w1t, w2t)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, act, pre = ctx.saved_tensors
        w1t, w2t = ctx.wts
        dy = _as(dy, h.dtype)
        dW2, db2 = _wgrad(dy, act, True)
        dpre = ops.gemm_nt(dy, w2t, L.EPI_MUL_DGELU, aux=pre)            # (dy W2) * GELU'(fc1 pre-activation)
        dW1, db1 = _wgrad(dpre, h, True)
        dh = ops.gemm_nt(dpre, w1t, L.EPI_STORE) if ctx.needs_input_grad[0] else None
        return dh, dW1, db1, dW2, db2, None


class SpatialAttnFn(torch.autograd.Function):
    """per (frame, head) softmax(q k^T d^-0.5) v over the P tokens (vision_transformer.py:206-214); qkv (F,P,3C) -> (F,P,C)"""

    @staticmethod
    def forward(ctx, qkv, H, impl):
        qkv = _as(qkv, qkv.dtype)
        o, lse = ops.attn_spatial_fwd(qkv, H, impl)
        ctx.save_for_backward(qkv, o, lse)
        ctx.H, ctx.impl = H, impl
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        return ops.attn_spatial_bwd(qkv, o, _as(d_o, qkv.dtype), lse, ctx.H, impl=ctx.impl), None, None


class TemporalAttnFn(torch.autograd.Function):
    """attention across the T frames of a clip at the same token position (vision_transformer.py:216-228)"""

    @staticmethod
    def forward(ctx, qkv, H, T):
        qkv = _as(qkv, qkv.dtype)
        if qkv.shape[0] % T:
            raise ValueError(f"{qkv.shape[0]} frames are not a whole number of clips of seqlen={T}")
        o, lse = ops.attn_temporal_fwd(qkv, H, T)
        ctx.save_for_backward(qkv, o, lse)
        ctx.H, ctx.T = H, T
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        return ops.attn_temporal_bwd(qkv, o, _as(d_o, qkv.dtype), lse, ctx.H, ctx.T), None, None


class TokenMeanFn(torch.autograd.Function):
    """x.mean(dim=1, keepdim=True) (vision_transformer.py:169) with fp32 accumulation"""

    @staticmethod
    def forward(ctx, x):
        Fr, P, C_ = x.shape
        ctx.P = P
        x = _as(x, x.dtype)
        return ops.st_colmean(x, x)[:, :C_].reshape(Fr, 1, C_)

    @staticmethod
    def backward(ctx, dm):
        return (dm / ctx.P).expand(-1, ctx.P, -1)


class StMixFn(torch.autograd.Function):
    """attentive addition of the parallel mode (vision_transformer.py:152-158): token means of [x_s || x_t] -> ts_attn Linear ->
    pairwise softmax -> blend.  x_s, x_t (F,P,C) compute dtype -> mix (F,P,C) compute dtype"""

    @staticmethod
    def forward(ctx, x_s, x_t, w_ts, b_ts, cache):
        (wc, wt), = cache.get([w_ts], x_s.dtype)
        x_s, x_t = _as(x_s, x_s.dtype), _as(x_t, x_s.dtype)
        means = ops.st_colmean(x_s, x_t)
        logits = ops.gemm_nt(means, wc, L.EPI_STORE_F32, bias=b_ts)
        ctx.save_for_backward(x_s, x_t, means, logits)
        ctx.wt = wt
        return ops.st_mix_fwd(x_s, x_t, logits)

    @staticmethod
    def backward(ctx, dmix):
        x_s, x_t, means, logits = ctx.saved_tensors
        dx_s, dx_t, dlogits = ops.st_mix_bwd(_as(dmix, x_s.dtype), x_s, x_t, logits, lambda dl: ops.gemm_nt(dl, ctx.wt, L.EPI_STORE))
        dW, db = _wgrad(dlogits, means, True)
        return dx_s, dx_t, dW, db, None


class TanhLinearFn(_ApplyMixin, torch.autograd.Function):
    """tanh(x W^T + b) (pre_logits, vision_transformer.py:350-353): the GEMM's TANH epilogue forward, maed_tanh_bwd + the two GEMMs backward"""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, direct=False):
        (wc, wt), = cache.get([weight], x.dtype)
        x = _as(x, x.dtype)
        y = ops.gemm_nt(x, wc, L.EPI_TANH, bias=bias)
        ctx.save_for_backward(x, y)
        ctx.wt = wt
        ctx.direct = _direct_begin(ctx, direct, (1, 2), weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        g = torch.empty_like(y)
        dy32 = _as(dy, torch.float32)        # (a named temporary: its storage must outlive the call that reads it)
        L.check(L.lib().maed_tanh_bwd(ops._p(dy32), ops._p(y), ops._p(g), y.numel(), ops.dt_code(y.dtype), ops._stream()), "tanh_bwd")
        dx = ops.gemm_nt(g, ctx.wt, L.EPI_STORE) if ctx.needs_input_grad[0] else None
        if ctx.direct is not None and _wgrad_direct(g, x, ctx.direct):
            return dx, None, None, None, None
        dW, db = _wgrad(g, x, True)
        return dx, dW, db, None, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in training on fp32 rows (ktd.py:54,56); the mask is a hash of (seed, index) -- recomputed, not stored"""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _as(x, torch.float32)
        y = torch.empty_like(x)
        L.check(L.lib().maed_dropout(ops._p(x), ops._p(y), x.numel(), p, seed, ops._stream()), "dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _as(dy, torch.float32)
        dx = torch.empty_like(dy)
        L.check(L.lib().maed_dropout(ops._p(dy), ops._p(dx), dy.numel(), ctx.p, ctx.seed, ops._stream()), "dropout(bwd)")
        return dx, None, None


class DropoutDevFn(torch.autograd.Function):
    """DropoutFn with the seed read from the device record of the running step (ops.DeviceTrainState): the launch arguments do not change from step to step"""

    @staticmethod
    def forward(ctx, x, p, state, call_id):
        x = _as(x, torch.float32)
        y = torch.empty_like(x)
        L.check(L.lib().maed_dropout_dev(ops._p(x), ops._p(y), x.numel(), p, ops._p(state), call_id, ops._stream()), "dropout_dev")
        ctx.p, ctx.state, ctx.call_id = p, state, call_id
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _as(dy, torch.float32)
        dx = torch.empty_like(dy)
        L.check(L.lib().maed_dropout_dev(ops._p(dy), ops._p(dx), dy.numel(), ctx.p, ops._p(ctx.state), ctx.call_id, ops._stream()), "dropout_dev(bwd)")
        return dx, None, None, None


def dropout(x, p, training):
    """F.dropout on the library: identity in eval / p = 0; the seed comes from torch's CPU generator (torch.manual_seed reproduces a run)"""
    if not training or p == 0.0:
        return x
    st = ops.DEVICE_STATE
    if st is not None and st.dev.device == x.device:
        return DropoutDevFn.apply(x, float(p), st.dev, st.next_call_id())
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return DropoutFn.apply(x, float(p), seed)


def attention_parallel(attn, h, seqlen, compute_dtype, impl):
    """stand-alone DIFFERENTIABLE Attention.forward in 'parallel' mode (vision_transformer.py:146-158,176) from the staged Functions
    (inside a Block the same arithmetic runs as one fused call per direction): h (F,P,C) -> fp32 (F,P,C)"""
    Fr, P, C_ = h.shape
    h = h.to(compute_dtype)
    qkv = LinearTokFn.apply(h.reshape(-1, C_), attn.qkv.weight, attn.qkv.bias, attn._cache, False).view(Fr, P, 3 * C_)
    x_s = SpatialAttnFn.apply(qkv, attn.num_heads, impl)
    x_t = TemporalAttnFn.apply(qkv, attn.num_heads, seqlen)
    mix = StMixFn.apply(x_s, x_t, attn.ts_attn.weight, attn.ts_attn.bias, attn._cache)
    return LinearTokFn.apply(mix.reshape(-1, C_), attn.proj.weight, attn.proj.bias, attn._cache, True).view(Fr, P, C_)


def attention(attn, h, seqlen, compute_dtype, impl):
    """Attention.forward for mode != 'parallel': h (F,P,C) compute dtype -> fp32 (F,P,C), or (F,1,C) in 'temporal' mode"""
    Fr, P, C_ = h.shape
    H, mode = attn.num_heads, attn.mode
    cache = attn._cache

    def qkv_of(t):
        return LinearTokFn.apply(t.reshape(-1, C_), attn.qkv.weight, attn.qkv.bias, cache, False).view(t.shape[0], t.shape[1], 3 * C_)

    if mode == 'series':          # :139-145  (the SAME qkv projection before each stage)
        o = SpatialAttnFn.apply(qkv_of(h), H, impl)
        o = TemporalAttnFn.apply(qkv_of(o), H, seqlen)
    elif mode == 'vanilla':       # :164-167
        o = SpatialAttnFn.apply(qkv_of(h), H, impl)
    elif mode == 'temporal':      # :168-173
        o = TemporalAttnFn.apply(qkv_of(TokenMeanFn.apply(h)), H, seqlen)
    elif mode == 'coupling':      # :160-163, 180-204
        if Fr % seqlen:
            raise ValueError(f"{Fr} frames are not a whole number of clips of seqlen={seqlen}")
        o = SpatialAttnFn.apply(qkv_of(h).view(Fr // seqlen, seqlen * P, 3 * C_), H, impl).view(Fr, P, C_)
    else:
        raise NotImplementedError(mode)
    out = LinearTokFn.apply(o.reshape(-1, C_), attn.proj.weight, attn.proj.bias, cache, True)
    return out.view(Fr, o.shape[1], C_)


def block(blk, x, seqlen):
    """Block.forward (vision_transformer.py:258-261) for mode != 'parallel'; x fp32 (F,P,C)"""
    Fr, P, C_ = x.shape
    cd = blk.compute_dtype
    x = x.float()
    h = LayerNormFn.apply(x.reshape(-1, C_), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, cd).view(Fr, P, C_)
    x = x + attention(blk.attn, h, seqlen, cd, blk.impl)          # 'temporal': (F,1,C) broadcasts over the tokens
    h = LayerNormFn.apply(x.reshape(-1, C_), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, cd)
    m = MlpFn.apply(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.mlp._cache)
    return x + m.view(Fr, P, C_)
