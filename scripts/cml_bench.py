import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
ctx = ops.get_context(0); dev = ctx.device
U, I, F, B = 1000000, 100000, 128, 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
Gu = ((torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
Gi = ((torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
Bi = ((torch.rand(I, generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, seed=5)
pos = ops.DeviceCSR.from_tensors(ip, ix, I)
st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
ctx.timing(True)
for it in range(7):
    if it == 2:
        torch.cuda.synchronize(); ctx.timing_report(); t0 = time.perf_counter()
    u, i, j = ops.bpr_sample(ctx, pos, B, seed=3, first_sample=it * B)
    st.train_step(u, i, j, 0.001, 0.001, 0.001, 0.5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
for n, (c, ms) in sorted(ctx.timing_report().items(), key=lambda kv: -kv[1][1]):
    print(f"{n}: {ms / 5:.4f} ms/step")
print(f"CML: wall {dt * 1e3:.3f} ms/step -> {B / dt / 1e6:.1f} M triplets/s; loss {st.pop_loss():.4e}")
