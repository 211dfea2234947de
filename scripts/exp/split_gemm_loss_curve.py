"""The same NeuMF training run (bench leg's shape, 300 steps) with the Dense products on the fp32 matrix instruction and on the bf16
instruction with three-way split operands (EL_GEMM_SPLIT = 0 / 1): per-step losses side by side."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
ctx = ops.get_context(0); dev = ctx.device
U, I, F, B, STEPS = 1_250_000, 1_000_000, 128, 262_144, 300
ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, seed=5)
pos = ops.DeviceCSR.from_tensors(ip, ix, I)
def run(split):
    os.environ["EL_GEMM_SPLIT"] = split
    g = torch.Generator(device=dev); g.manual_seed(3)
    gu = lambda a, b: (torch.rand((a, b), generator=g, device=dev) * 2 - 1) * (6.0 / (a + b)) ** 0.5
    units = [4 * F, 2 * F, F]
    w = {"Umf": gu(U, F), "Imf": gu(I, F), "Umlp": gu(U, F), "Imlp": gu(I, F), "W": [], "b": []}
    kin = 2 * F
    for n_out in units:
        w["W"].append(gu(kin, n_out)); w["b"].append(torch.zeros(n_out, device=dev)); kin = n_out
    w["hw"], w["hb"] = gu(F + units[-1], 1)[:, 0].contiguous(), torch.zeros(1, device=dev)
    st = ops.NmfDeviceState(ctx, w, max_batch=B)
    out = []
    for s in range(STEPS):
        u, i, y = ops.pointwise_sample(ctx, pos, B, seed=3, first_sample=s * B)
        st.train_step(u, i, y, 0.001)
        out.append(st.pop_loss())
    del st
    torch.cuda.empty_cache()
    return np.array(out)
a, b = run("0"), run("1")
rel = np.abs(a - b) / np.abs(a)
for s in (0, 1, 10, 50, 100, 200, 299):
    print(f"step {s + 1}: loss fp32-instruction {a[s]:.9f}  split {b[s]:.9f}  rel diff {rel[s]:.2e}")
a2 = run("0")
rel2 = np.abs(a - a2) / np.abs(a)
b2 = run("1")
rel3 = np.abs(b - b2) / np.abs(b)
print(f"max relative difference over {STEPS} steps: split vs fp32 instruction {rel.max():.2e} (step {int(rel.argmax()) + 1}); fp32 instruction vs a second run of "
      f"itself {rel2.max():.2e} (step {int(rel2.argmax()) + 1}; step 2: {rel2[1]:.2e}, step 11: {rel2[10]:.2e}, step 101: {rel2[100]:.2e}); split vs a second run of itself {rel3.max():.2e}")
