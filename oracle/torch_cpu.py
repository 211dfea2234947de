"""Multi-threaded torch-CPU restatement of the reference's TensorFlow path.  TEST / BENCH INFRASTRUCTURE ONLY
(imported by bench.py's `cpu_baseline` leg and by tests/; never by the product package).

"parity unpinned": TensorFlow 2.3.2 is not installable here, so this restates -- it does not execute --
elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch_model.py:
  :49-55  call        xui = Bi[item] + sum(Gu[user] * Gi[item], 1)
  :58-80  train_step  softplus loss + l2 terms, gradients (IndexedSlices, duplicates summed), Keras Adam whose sparse apply
                      decays m, v and moves theta for EVERY row
  :83-88  predict / get_top_k   matmul + where(mask, preds, -inf) + top_k(sorted=True)
It is the "TF eager on the host cores" stand-in of SURVEY.md 8(d)-(2): BLAS GEMM + torch ops on
`torch.set_num_threads(os.cpu_count())` threads, fp32.  tests/test_oracle_torch_cpu.py pins it to the NumPy
restatement (oracle/bprmf_batch.py) and the C top-k oracle.
"""
import os

import torch

BETA1, BETA2, EPS = 0.9, 0.999, 1e-7


def usable_cores():
    """Cores this process may actually run on: min(os.cpu_count(), CPU affinity mask, cgroup CPU quota).  A container on a
    256-thread host with an 8-CPU quota reports os.cpu_count() == 256; 256 OpenMP threads on 8 CPUs run slower than one."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0]) if txt[0] not in ("max", "-1") else -1.0
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def use_all_cores(n=None):
    torch.set_num_threads(n or usable_cores())
    return torch.get_num_threads()


class BprmfBatchTorchCpu:
    """State + one train_step of BPRMF_batch on the CPU (fp32 tensors, in place)."""

    def __init__(self, Gu, Gi, Bi, lr=0.001, l_w=0.1, l_b=0.001):
        t = lambda x: torch.as_tensor(x, dtype=torch.float32).clone()
        self.Gu, self.Gi, self.Bi = t(Gu), t(Gi), t(Bi)
        self.m = [torch.zeros_like(x) for x in (self.Gu, self.Gi, self.Bi)]
        self.v = [torch.zeros_like(x) for x in (self.Gu, self.Gi, self.Bi)]
        self.lr, self.l_w, self.l_b = float(lr), float(l_w), float(l_b)
        self.step = 0

    def train_step(self, u, i, j):
        """u, i, j: int64 tensors [B].  Returns the batch loss (python float)."""
        Gu, Gi, Bi = self.Gu, self.Gi, self.Bi
        gu, gi, gj = Gu[u], Gi[i], Gi[j]                               # embedding_lookup :49-51
        bi, bj = Bi[i], Bi[j]
        xui = bi + (gu * gi).sum(1)
        xuj = bj + (gu * gj).sum(1)
        d = xui - xuj
        dc = d.clamp(-80.0, 1e8)
        l2 = lambda x: 0.5 * (x * x).sum()
        loss = torch.nn.functional.softplus(-dc).sum() + self.l_w * (l2(gu) + l2(gi) + l2(gj)) + self.l_b * l2(bi) + \
            self.l_b * l2(bj) / 10.0
        s = torch.where(d >= -80.0, -torch.sigmoid(-d), torch.zeros_like(d))   # d loss / d difference (clip passes inside)
        # IndexedSlices: one gradient row per batch entry; OptimizerV2 sums duplicate indices (unsorted_segment_sum over the
        # unique indices) before the sparse apply
        uu, inv_u = torch.unique(u, return_inverse=True)
        gU = torch.zeros((uu.numel(), Gu.shape[1])).index_add_(0, inv_u, s[:, None] * (gi - gj) + self.l_w * gu)
        ij = torch.cat([i, j])
        ii, inv_i = torch.unique(ij, return_inverse=True)
        gI = torch.zeros((ii.numel(), Gi.shape[1])).index_add_(0, inv_i, torch.cat([s[:, None] * gu + self.l_w * gi,
                                                                                 -s[:, None] * gu + self.l_w * gj]))
        gB = torch.zeros(ii.numel()).index_add_(0, inv_i, torch.cat([s + self.l_b * bi, -s + (self.l_b / 10.0) * bj]))
        self.step += 1
        t = self.step
        lr_t = self.lr * (1.0 - BETA2 ** t) ** 0.5 / (1.0 - BETA1 ** t)
        # Keras Adam._resource_apply_sparse: m <- m b1 (ALL rows); m[idx] += (1-b1) g; v <- v b2 (ALL rows); v[idx] += (1-b2) g^2;
        # theta <- theta - lr_t m / (sqrt(v) + eps) (ALL rows)
        for th, idx, g, m, v in ((Gu, uu, gU, self.m[0], self.v[0]), (Gi, ii, gI, self.m[1], self.v[1]), (Bi, ii, gB, self.m[2], self.v[2])):
            m.mul_(BETA1)
            m.index_add_(0, idx, g, alpha=1.0 - BETA1)
            v.mul_(BETA2)
            v.index_add_(0, idx, g * g, alpha=1.0 - BETA2)
            th.addcdiv_(m, v.sqrt().add_(EPS), value=-lr_t)
        return float(loss)


def predict_topk(Gu_block, Gi, Bi, excl_indptr, excl_indices, k):
    """preds = Bi + Gu_block @ Gi^T; top_k(where(mask, preds, -inf), k) with mask = NOT in the user's train row.
    excl_indptr: int64 [n+1] rebased to the block, excl_indices: int64 [nnz]."""
    preds = torch.addmm(Bi[None, :], Gu_block, Gi.t())
    counts = excl_indptr[1:] - excl_indptr[:-1]
    rows = torch.repeat_interleave(torch.arange(Gu_block.shape[0]), counts)
    preds[rows, excl_indices] = float("-inf")
    v, idx = torch.topk(preds, k, dim=1, sorted=True)
    return idx, v
