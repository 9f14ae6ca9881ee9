"""One training step -- zero_grad, forward, loss, backward, gradient readiness, Adam -- captured ONCE as a hipGraph and replayed.

The reference's loop (lib/core/trainer.py:240-248: `optimizer.zero_grad(); loss.backward(); optimizer.step()` per batch) costs this framework ~11 ms of host time per
cfg3 step for ~550 launches on three streams; a replay costs the host one 32-byte copy and one graph launch.  The GPU side is unchanged (the same kernels in the same
stream structure: the library's side streams and the Python side stream are forked from and joined back into the capture through their event fences).

What makes the step replayable:
  * nothing that changes from step to step is a launch argument -- learning rate, Adam's bias corrections and the Dropout seed live in a device record
    (`ops.DeviceTrainState` = include/maed_hip.h `maed_train_state`) that the host rewrites before every replay (`maed_adam_step_dev`, `maed_dropout_dev`);
  * the batch is copied into static input tensors; the loss is a static output tensor;
  * the persistent K-stream GEMMs are switched off for the graph's lifetime (their slab hand-over protocol numbers LAUNCHES on the host: a replayed launch would
    carry a stale epoch) -- they are only taken from K >= 2560 (cfg5's MLP) anyway;
  * gradient all-reduces are not captured: one rank, or no graph (`GraphedTrainStep` raises when the bucketer runs collectives).

`eager=True` runs the very same step (same entry points, same device record) without capturing: the comparison arm of tests/test_gpu_graph.py and bench.py.
"""
import torch

from . import _lib as L
from . import ops


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, clip, target, warmup=2, eager=False, n_graphs=1):
        """n_graphs: executable graphs captured (identical steps over the same parameters, optimizer state and static inputs; each with its own activation pool) and
        replayed in turn.  Measured (profiles/r06_graph_overlap.txt): a replayed step keeps the three-stream concurrency but leaves the GPU idle for ~1 ms between
        replays on this runtime, and two or three graphs replayed in turn change nothing about that (21.0 ms against 20.0 eager, same box) -- the default is one."""
        dev = clip.device
        if getattr(optimizer, "bucketer", None) is not None and optimizer.bucketer.collectives:
            raise RuntimeError("GraphedTrainStep: gradient all-reduces are not captured (one rank only)")
        self.model, self.criterion, self.opt, self.eager = model, criterion, optimizer, eager
        self.clip = clip.clone()
        self.target = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in target.items()} if isinstance(target, dict) else target
        self.state = ops.DeviceTrainState(dev)
        optimizer.device_state = self.state
        self._prev_state = ops.DEVICE_STATE
        ops.DEVICE_STATE = self.state
        self._opts = (L.get_option(L.OPT_SK), L.get_option(L.OPT_TN_SK))
        L.set_option(L.OPT_SK, 0)
        L.set_option(L.OPT_TN_SK, 0)
        self.graph, self.loss = None, None
        self._graphs, self._losses, self._turn = [], [], 0
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        if eager or dev.type != "cuda":
            return
        # lazy initialisation (weight caches, kernel attributes, the allocator's pool, the side streams) happens in eager steps, on the stream the capture will use
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._prepare()
                self._body()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        self._keep = []                      # pointer tables the captured launches read
        for _ in range(max(1, n_graphs)):
            g = torch.cuda.CUDAGraph()
            ops._CAPTURE_KEEP = self._keep
            self._prepare(seed=0)            # (nothing executes during the capture; no draw from the host generator: the eager arm of a comparison makes none here)
            try:
                # "relaxed": pinned staging blocks may be allocated while capturing (hipHostMalloc is one of the calls the stricter modes refuse)
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="relaxed"):
                    self.state.calls = 0
                    loss = self._body()
            finally:
                ops._CAPTURE_KEEP = None
            # the capture advanced the optimizer's host-side step counter once without running a step
            self.opt.step_count -= 1
            self._graphs.append(g)
            self._losses.append(loss)
        self.graph, self.loss = self._graphs[0], self._losses[0]

    # ---- one step ---------------------------------------------------------------------------------------------------------------------------------------
    def _prepare(self, seed=None):
        """host side of a step: new Dropout seed, the hyper-parameters of the step about to run, one upload"""
        st, opt = self.state, self.opt
        g = opt.param_groups[0]
        t = opt.step_count + 1
        st.begin_step(seed)
        st.set_hyper(g["lr"], 1.0 - g["betas"][0] ** t, 1.0 - g["betas"][1] ** t)
        st.upload()

    def _body(self):
        self.opt.zero_grad()
        loss, _ = self.criterion(self.model(self.clip), self.target, None)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, clip=None, target=None):
        if clip is not None:
            self.clip.copy_(clip, non_blocking=True)
        if target is not None:
            for k, v in target.items():
                if torch.is_tensor(v):
                    self.target[k].copy_(v, non_blocking=True)
        self._prepare()
        if self.graph is None:
            return self._body()
        k = self._turn % len(self._graphs)
        self._turn += 1
        self._graphs[k].replay()
        self.opt.step_count += 1
        ops.bump_weight_epoch()          # the parameters changed behind the host's back: an eager forward after this must refresh its compute-dtype weight copies
        self.loss = self._losses[k]
        return self.loss

    def close(self):
        """hand the optimizer and the Dropout layers back to their host-side scalars"""
        self.opt.device_state = None
        ops.DEVICE_STATE = self._prev_state
        L.set_option(L.OPT_SK, self._opts[0])
        L.set_option(L.OPT_TN_SK, self._opts[1])
        self.graph = None
        self._graphs, self._losses = [], []
