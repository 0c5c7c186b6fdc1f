"""Host-side cost of one eager training step (python + autograd + ctypes launch path): cProfile over a few steps at a size where the GPU is never the limit."""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
cfg, _ = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "autopet96"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = cfg["input_size"][0]
torch.manual_seed(0)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, (B, 2, S, S, S), use_graph=False, overlap=False)
x, lab = synth(cfg, B, "cuda", 1)
from veloxseg_amd import functional as VF
VF.BRANCH_STREAMS = os.environ.get("VX_BRANCH_STREAMS", "1") == "1"
VF.USE_COMPOSITE = os.environ.get("VX_COMPOSITE", "1") == "1"
for _ in range(3):
    eng.step(x, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host ms/step (launch only) %.2f, incl. final sync %.2f" % ((t1 - t0) * 100, (t2 - t0) * 100))
if os.environ.get("VX_NO_CPROFILE") == "1":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    eng.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
