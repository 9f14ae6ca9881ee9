"""Data-parallel training runtime for one node of MI355X (reference: train.py:113,182 DDP over NCCL;
lib/utils/utils.py:127-132 Adam; lib/core/trainer.py:240-248 backward/step).

MI355X-first layout instead of torch DDP + per-tensor Adam:
  * every trainable parameter lives in ONE flat fp32 arena (params are views), gradients in a second
    arena with the same offsets, Adam moments in two more -> the optimizer step is one kernel
    (maed_adam_step) over 288 GB-class HBM-resident arenas, and a gradient bucket is a contiguous slice.
  * arena order follows the forward pass, so backward completes buckets from the END of the arena;
    as soon as every gradient of a bucket is final (autograd post-accumulate hooks for ATen-managed
    parameters, the fused Block backward's `grads_ready` callback for the STE) the bucket's
    all-reduce is launched asynchronously: RCCL runs it on its own HIP stream, fenced by events against
    the backward stream, i.e. overlapped with the remaining backward kernels.  xGMI is a
    point-to-point mesh (7 links/GPU), so buckets are large (32 MiB default) -- few, big collectives.
  * the division by world size is folded into the Adam kernel (gscale).
One process per GPU; rank/world come from torch.distributed (backend "nccl" == RCCL on ROCm).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops

ALIGN = 64  # elements: every view starts 256-B aligned


def _forward_order_key(name):
    """backbone -> proj -> embeddings -> blocks 0..n -> norm -> pre_logits -> decoder"""
    if name.startswith("encoder.patch_embed.backbone."):
        return (0, 0)
    if name.startswith("encoder.patch_embed."):
        return (1, 0)
    if name in ("encoder.cls_token", "encoder.pos_embed", "encoder.temp_embed"):
        return (2, 0)
    if name.startswith("encoder.blocks."):
        return (3, int(name.split(".")[2]))
    if name.startswith("encoder."):
        return (4, 0)
    return (5, 0)


class ParamArena:
    """Re-homes a model's trainable parameters (and their .grad) into flat fp32 arenas."""

    def __init__(self, model, device=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        definition_order = [id(p) for _, p in named]            # model.named_parameters() order: the reference optimizer's group order
        named.sort(key=lambda np_: _forward_order_key(np_[0]))  # stable: keeps definition order inside a group
        device = device or named[0][1].device
        self.names, self.params, self.offsets = [], [], []
        off = 0
        for n, p in named:
            self.names.append(n)
            self.params.append(p)
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.definition_order = [self.index[i] for i in definition_order]

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class RcclComm:
    """The library's own communicator (maed_comm_*: RCCL bound from the copy PyTorch-ROCm ships, side HIP stream, event
    fences) instead of torch.distributed's ProcessGroupNCCL.  The 128-byte unique id travels from rank 0 through the
    already-initialised torch.distributed group (any backend) -- or no exchange at all for world == 1."""

    @staticmethod
    def _lib_path(lib_path=None):
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return lib_path or os.environ.get("MAED_RCCL_LIB") or (bundled if os.path.exists(bundled) else None)

    @staticmethod
    def available(lib_path=None):
        """NON-collective feasibility check (every rank may call it independently): the NCCL-API library loads and exports what maed_comm_* binds.  Returns (ok, why).
        Creating the communicator itself is collective (unique-id broadcast + ncclCommInitRank): ranks must agree on feasibility FIRST -- a rank that failed here while
        its peers sat inside the constructor would hang the job instead of falling back."""
        try:
            path = RcclComm._lib_path(lib_path)
            L.check(L.lib().maed_comm_load(path.encode() if path else None), "comm_load")
            return True, ""
        except Exception as e:  # noqa: BLE001
            return False, str(e)

    def __init__(self, rank=None, world=None, lib_path=None):
        """lib_path: the NCCL-API library to bind (default: $MAED_RCCL_LIB, else the librccl.so PyTorch-ROCm ships -- one RCCL per process -- else the loader's
        librccl.so.1).  tests/test_comm_world2.py names a shared-memory stand-in here to run two ranks on a box without GPUs."""
        lib = L.lib()
        have_pg = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank() if have_pg else 0)
        self.world = world if world is not None else (dist.get_world_size() if have_pg else 1)
        path = RcclComm._lib_path(lib_path)
        L.check(lib.maed_comm_load(path.encode() if path else None), "comm_load")
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            L.check(lib.maed_comm_unique_id(uid), "comm_unique_id")
        if self.world > 1:
            box = [uid.raw]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(box[0], 128)
        L.check(lib.maed_comm_init(self.rank, self.world, uid), "comm_init")

    def allreduce_async(self, t):
        L.check(L.lib().maed_comm_allreduce_async(ops._p(t), t.numel(), ops.dt_code(t.dtype), ops._stream()), "comm_allreduce_async")

    def wait(self):
        L.check(L.lib().maed_comm_wait(ops._stream()), "comm_wait")

    def destroy(self):
        L.check(L.lib().maed_comm_destroy(), "comm_destroy")


class GradBucketer:
    """Bucketed, overlapped gradient all-reduce over the gradient arena.  comm=None: torch.distributed (backend "nccl" is
    RCCL); comm=RcclComm(): the library's own RCCL communicator and side stream (MAED_COMM=direct in bench.py)."""

    def __init__(self, arena, model, bucket_bytes=32 << 20, process_group=None, force_collectives=False, comm=None):
        self.arena = arena
        self.pg = process_group
        self.comm = comm
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if comm is not None:
            self.world = comm.world
        # world == 1 normally skips the collectives; force_collectives issues them anyway (exercises the RCCL
        # stream/event plumbing on a single GPU: tests/test_gpu_model.py)
        self.collectives = (self.world > 1) or (force_collectives and (comm is not None or (dist.is_available() and dist.is_initialized())))
        cap = max(1, bucket_bytes // 4)
        # Bucket boundaries also fall where a group of kernel-written gradients that is reported as ONE unit begins (the backbone's per-stage
        # weight-standardisation groups: last stage first, stem + first stage last): a bucket then never waits for a LATER report than its own group's,
        # and what remains exposed at the end of the backward is the all-reduce of the small stem + stage-1 group (0.9 MB at cfg3), not of a 32 MiB
        # bucket that happens to contain it.
        cuts = set()
        if self.world > 1 or force_collectives:
            for m in model.modules():
                for g in getattr(m, "_ws_groups", ()):
                    idx = [arena.index[id(p)] for p in g.fused_parameters() if id(p) in arena.index]
                    if idx:
                        cuts.add(min(idx))
        # buckets are built from the END of the arena (first to complete in backward)
        self.buckets = []  # [start, end, n_params]
        self.bucket_of = [0] * len(arena.params)
        end = arena.numel
        cur_start, cur_n = end, 0
        for i in range(len(arena.params) - 1, -1, -1):
            o = arena.offsets[i]
            if cur_n > 0 and end - o > cap:
                self.buckets.append([cur_start, end, cur_n])
                end, cur_n = cur_start, 0
            cur_start = o
            cur_n += 1
            self.bucket_of[i] = len(self.buckets)
            if i in cuts and i > 0:             # first parameter of a reporting group: the parameters in front of it belong to an earlier-in-forward group
                self.buckets.append([cur_start, end, cur_n])
                end, cur_n = cur_start, 0
        if cur_n:
            self.buckets.append([cur_start, end, cur_n])
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen = [False] * len(arena.params)      # parameters that reported a gradient since the last finish()
        self.unmarked = []                             # ... and the ones that did not, as of the last finish() (FusedAdam skips them like torch.optim.Adam)
        self._works = []
        self.launch_order = []                         # bucket indices in the order they were launched in the last step (diagnostics: bench.py "ddp")
        self._finished = False                         # finish() ran and nothing was reported since: a second finish() is a no-op
        self._fused = set()
        self._fused_modules = []
        for m in model.modules():  # modules whose kernels write parameter gradients directly (STE Block, ResNetV2)
            if hasattr(m, "fused_parameters") and hasattr(m, "grads_ready"):
                m.grads_ready = self._block_ready
                self._fused_modules.append(m)
                self._fused.update(id(p) for p in m.fused_parameters())
        for p in arena.params:
            if id(p) not in self._fused:
                p.register_post_accumulate_grad_hook(self._param_ready)
                p._maed_ready = self._param_ready       # a backward kernel that accumulates into .grad itself (ops.direct_grad_slot) reports through this

    # ---- readiness -------------------------------------------------------------------------------
    def _mark(self, p):
        i = self.arena.index.get(id(p))
        if i is None or self._seen[i]:      # (a parameter is final once per step, whoever says so first)
            return
        self._seen[i] = True
        self._finished = False
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _param_ready(self, p):
        self._mark(p)

    def _block_ready(self, block):
        for p in block.fused_parameters():
            self._mark(p)

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        if not any(self._launched[i] for i in range(len(self._launched)) if i != b):
            self.launch_order = []
        self.launch_order.append(b)
        if self.collectives:
            s, e, _ = self.buckets[b]
            if self.comm is not None:
                self.comm.allreduce_async(self.arena.grad[s:e])
            else:
                self._works.append(dist.all_reduce(self.arena.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Launch whatever has not fired (parameters without a gradient this step), then make the
        current stream wait for every bucket.  Once per step, before the optimizer -- FusedAdam.step() calls it; calling it yourself first (to clip or
        inspect the reduced gradients) is fine: until a new gradient is reported or the optimizer has consumed the step, further calls change nothing."""
        if self._finished:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        if self.comm is not None and self.collectives:
            self.comm.wait()
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        # kernel-written gradients are reported by the module's LAST backward kernel (ResNetV2: the weight-standardisation backward).  A module that ran a
        # grad-enabled forward but whose reporter cannot run -- a backbone with frozen convolution weights and trainable GroupNorm parameters -- still
        # produced gradients: count its trainable parameters as seen instead of freezing them
        for m in self._fused_modules:
            if getattr(m, "_grad_forward_seen", False):
                for p in m.fused_parameters():
                    i = self.arena.index.get(id(p))
                    if i is not None:
                        self._seen[i] = True
                m._grad_forward_seen = False
        self.unmarked = [i for i, seen in enumerate(self._seen) if not seen]
        self._seen = [False] * len(self._seen)
        for p in self.arena.params:         # forwards counted in by ops.direct_grad_begin whose backward never ran must not hold the next step's report back
            if getattr(p, "_maed_direct", 0):
                p._maed_direct = 0
        for m in self._fused_modules:       # a backward that never ran (an exception, a detached output) must not poison the next step
            m._pending_backwards = 0
            for g in getattr(m, "_ws_groups", ()):      # per-stage weight standardisation: the stage owners count their own backwards
                g._pending_backwards = 0
        self._finished = True

    def consumed(self):
        """the optimizer has applied this step: the next finish() evaluates afresh (also when no gradient at all is reported before it)"""
        self._finished = False

    def broadcast_parameters(self, src=0):
        """train.py:113 DDP construction broadcasts rank 0's parameters once."""
        if self.collectives and dist.is_available() and dist.is_initialized():
            dist.broadcast(self.arena.flat, src=src, group=self.pg)
            ops.bump_weight_epoch()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay) as ONE kernel over the arena (maed_adam_step).

    Drop-in for the optimizer `lib/utils/utils.py:127-132` builds: a torch.optim.Optimizer with one parameter group per
    tensor in `model.named_parameters()` order (pass `model`), so `LambdaLR` (train.py:123-127) drives `param_groups[*]['lr']`
    and `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format -- the 'optimizer' entry of the reference's
    `epoch_N.pth.tar` checkpoints (trainer.py:330-368) resumes here and vice versa.  Compute-dtype weight copies are
    rebuilt lazily by the modules after a step (ops.WEIGHT_EPOCH)."""

    def __init__(self, arena, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, bucketer=None, model=None):
        self.arena, self.bucketer = arena, bucketer
        order = list(getattr(arena, "definition_order", range(len(arena.params))))     # named_parameters() order even without model=
        if model is not None:   # the reference's group order = named_parameters() order (the arena is in forward order)
            order = [arena.index[id(p)] for _, p in model.named_parameters() if id(p) in arena.index]
            assert len(order) == len(arena.params)
        self._order = order
        groups = [{"params": [arena.params[i]], "name": arena.names[i]} for i in order]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.step_count = 0
        self.device_state = None    # ops.DeviceTrainState: lr and bias corrections are then read from device memory (graphed.GraphedTrainStep; uniform groups only)
        self._stepped = set()       # arena indices that have received at least one update (torch.optim.Adam keeps no state for the others)

    # kept as attributes for callers that read them
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _uniform(self):
        g0 = self.param_groups[0]
        key = (g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"])
        return all((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) == key for g in self.param_groups)

    @torch.no_grad()
    def step(self, closure=None):
        world = 1
        if self.bucketer is not None:
            self.bucketer.finish()
            world = self.bucketer.world
        self.step_count += 1
        a = self.arena
        # torch.optim.Adam skips parameters whose .grad is None (ts_attn in the non-parallel st_modes, any unused parameter): no moment decay, no weight
        # decay.  The arena's gradients are never None, so "received no gradient this step" comes from the bucketer's readiness reports.
        skip = set(self.bucketer.unmarked) if self.bucketer is not None else set()
        if self.bucketer is not None:
            self.bucketer.consumed()
        self._stepped.update(i for i in range(len(a.params)) if i not in skip)
        if self._uniform():     # the reference's case: every group shares the schedule -> one launch over the whole arena (or per run of active tensors)
            g = self.param_groups[0]
            runs = [(0, a.numel)]
            if skip:
                runs, start = [], None
                for i in range(len(a.params)):
                    end_i = a.offsets[i + 1] if i + 1 < len(a.params) else a.numel
                    if i in skip:
                        if start is not None:
                            runs.append((start, a.offsets[i])); start = None
                    elif start is None:
                        start = a.offsets[i]
                    if i + 1 == len(a.params) and start is not None:
                        runs.append((start, end_i))
            st = self.device_state
            if st is not None:
                # the record is uploaded here unless the step is being captured: a replay's record is written by the replaying host code (graphed.py)
                st.set_hyper(g["lr"], 1.0 - g["betas"][0] ** self.step_count, 1.0 - g["betas"][1] ** self.step_count)
                if not (a.flat.is_cuda and torch.cuda.is_current_stream_capturing()):
                    st.upload()
            for lo, hi in runs:
                if st is not None:
                    ops.adam_step_dev(a.flat[lo:hi], a.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], None, st.dev, g["betas"][0], g["betas"][1],
                                      g["eps"], g["weight_decay"], gscale=1.0 / world)
                else:
                    ops.adam_step(a.flat[lo:hi], a.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], None, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                  g["weight_decay"], self.step_count, gscale=1.0 / world)
        else:                   # per-group hyper-parameters: one launch per tensor
            if self.device_state is not None:
                raise RuntimeError("FusedAdam.device_state needs one schedule for every parameter group (the reference's case)")
            for g, i in zip(self.param_groups, self._order):
                if i in skip:
                    continue
                o, n = a.offsets[i], a.params[i].numel()
                ops.adam_step(a.flat[o:o + n], a.grad[o:o + n], self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], None, g["lr"],
                              g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, gscale=1.0 / world)
        ops.bump_weight_epoch()

    # ---- torch.optim.Adam-compatible (de)serialisation ------------------------------------------------------------
    def _views(self, i):
        o, p = self.arena.offsets[i], self.arena.params[i]
        return self.exp_avg[o:o + p.numel()].view(p.shape), self.exp_avg_sq[o:o + p.numel()].view(p.shape)

    def state_dict(self):
        # (one step count for every tensor that has state: a parameter that starts receiving gradients late is bias-corrected with the global count, where
        # torch keeps a per-tensor count -- the reference's models have no such parameter)
        if self.step_count > 0:
            for i in self._order:
                if i not in self._stepped:      # never updated: no state, as torch.optim.Adam
                    continue
                m, v = self._views(i)
                self.state[self.arena.params[i]] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)     # validates group structure, casts state to the parameters' device/dtype
        steps = []
        for i in self._order:
            st = self.state.get(self.arena.params[i])
            if not st:
                continue
            m, v = self._views(i)
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps.append(int(float(st["step"])))
            self._stepped.add(i)
        if steps:
            assert len(set(steps)) == 1, "per-tensor step counts differ: not an Adam state this optimizer can represent"
            self.step_count = steps[0]
        self.state.clear()
