"""How the screened NeuMF scoring's bound holds up as the network trains: survivors (pairs the fp32 kernel re-scores) after 0 / 200 /
1000 / 3000 Adam steps at the bench leg's shape (1.25 M x 1 M, d = 128, B = 262 144, lr 0.001)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
ctx = ops.get_context(0); dev = ctx.device
U, I, F, B = 1_250_000, 1_000_000, 128, 262_144
ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, seed=5)
pos = ops.DeviceCSR.from_tensors(ip, ix, I)
g = torch.Generator(device=dev); g.manual_seed(3)
gu = lambda a, b: (torch.rand((a, b), generator=g, device=dev) * 2 - 1) * (6.0 / (a + b)) ** 0.5
units = [4 * F, 2 * F, F]
w = {"Umf": gu(U, F), "Imf": gu(I, F), "Umlp": gu(U, F), "Imlp": gu(I, F), "W": [], "b": []}
kin = 2 * F
for n_out in units:
    w["W"].append(gu(kin, n_out)); w["b"].append(torch.zeros(n_out, device=dev)); kin = n_out
w["hw"], w["hb"] = gu(F + units[-1], 1)[:, 0].contiguous(), torch.zeros(1, device=dev)
st = ops.NmfDeviceState(ctx, w, max_batch=B)
done = 0
for target in tuple(int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,200,1000,3000".split(","))):
    while done < target:
        u, i, y = ops.pointwise_sample(ctx, pos, B, seed=3, first_sample=done * B)
        st.train_step(u, i, y, 0.001); done += 1
    loss = st.pop_loss() / max(target, 1) if target else float("nan")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.score_topk_logits(0, 128, 12, excl=pos, screen=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pairs, fb = st.screen_stats()
    bn = [float(b.abs().max()) for b in st.b]
    print(f"steps {target}: exact pairs {pairs} = {pairs / (128 * I):.5f} of the block, fell back {fb}, {dt * 1e3:.1f} ms (first call incl. item image); "
          f"max |bias| per layer {bn}, mean loss so far {loss:.4f}", flush=True)
