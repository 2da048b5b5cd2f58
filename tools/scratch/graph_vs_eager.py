import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mini-sglang_amd"))
from harness import workloads as W
from mi355_attn import ops
dev = "cuda:0"
c = W.decode_case(dev, 32, 8, 64, W.cfg2_lengths(1), 16, ps=16, room=8)
for _ in range(2):
    W.decode_step(c)
us, nb = W.time_decode_launches(c, 20)
print("eager", round(us, 2), "us", round(nb / us / 1e3), "GB/s")
md = c.backend.forward_metadata
ws = torch.empty(16, dtype=torch.uint8, device=dev)
pools = [c.kv.get_kv_cache(l) for l in range(c.L)]
out = torch.empty((c.B, c.Hq * c.D), dtype=torch.bfloat16, device=dev)
def launches():
    for kp, vp in pools:
        ops.decode_attn_planned(c.q, kp, vp, md.plan, c.Hq, workspace=ws, out=out, k_new=c.k, v_new=c.v)
launches(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    launches()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record(); torch.cuda.synchronize()
us2 = e0.elapsed_time(e1) * 1e3 / (20 * c.L)
print("graph", round(us2, 2), "us", round(nb / us2 / 1e3), "GB/s")
