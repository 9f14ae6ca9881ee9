"""which part of the train step refuses to be captured?  progressively larger bodies under torch.cuda.graph (tiny model)"""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_graph import _setup
from maed_amd import ops, _lib as L
model, arena, opt, crit, clip, tgt = _setup()
st = ops.DeviceTrainState(clip.device); opt.device_state = st; ops.DEVICE_STATE = st
st.begin_step(1); st.set_hyper(1e-4, 0.1, 0.001); st.upload()
def fwd():
    with torch.no_grad():
        return model(clip)["theta"]
def fwd_grad():
    return model(clip)["theta"]
def fwd_loss():
    return crit(model(clip), tgt, None)[0]
def fwd_bwd():
    opt.zero_grad(); l = crit(model(clip), tgt, None)[0]; l.backward(); return l
def full():
    opt.zero_grad(); l = crit(model(clip), tgt, None)[0]; l.backward(); opt.step(); return l
s = torch.cuda.Stream()
for name, body in (("forward, no grad", fwd), ("forward with grad", fwd_grad), ("forward + loss", fwd_loss), ("forward + loss + backward", fwd_bwd), ("whole step", full)):
    try:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                st.calls = 0
                body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        keep = []
        ops._CAPTURE_KEEP = keep
        with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
            st.calls = 0
            out = body()
        ops._CAPTURE_KEEP = None
        g.replay(); torch.cuda.synchronize()
        print(f"{name}: captured and replayed, out {float(out.float().sum()):.4f}", flush=True)
    except Exception as e:
        print(f"{name}: FAILED {type(e).__name__}: {str(e).splitlines()[0]}", flush=True)
        traceback.print_exc()
        break
