"""Host-side (Python) profile of one eval forward: where the ~7 ms of enqueue time per forward go at the reference's contract shape (16 x 224 x 224),
which bounds the pipelined step there.  cProfile over 20 forwards, no GPU synchronisation inside; top functions by own and cumulative time."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_batch, trained_like_
torch.set_grad_enabled(False)
dev = torch.device("cuda"); B = int(os.environ.get("HP_B", 16)); S = int(os.environ.get("HP_S", 224)); amp = os.environ.get("HP_DTYPE", "bf16") == "bf16"
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=S)))).to(dev).eval()
b, _, _ = make_batch(B, 2048, 5000, S, seed=700, device=dev); b["coarse_rand"] = torch.rand(B, 18000, device=dev)
def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        return model(dict(b))
for _ in range(3): fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): fwd()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 20:.2f} ms per forward; with the final drain {1e3 * (t2 - t0) / 20:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): fwd()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime")
st.print_stats(28)
st.sort_stats("cumulative"); st.print_stats(30)
