"""The software pipeline of the single-GPU BPR-MF training loop (BPRMF_batch.py:100-109: `for batch in sampler.step(...):
train_step(batch)`): the sampler never reads the model, so the batch of step t+1 is drawn -- and ordered for the segment kernels --
on a side stream while step t's kernels run.  bench.py's headline leg and tests/test_gpu_fullsize_c4.py drive exactly this code.
"""
import torch

from . import ops


class PrefetchSampler:
    """BPR triplets of step t+1 are drawn on a side stream while step t trains: the sampler (custom_sampler.py:31-46) does not
    depend on the model, so a training loop can always run it one batch ahead.  Two triplet buffers, events both ways."""

    def __init__(self, ctx, pos, B, seed, enabled=True, presort_state=None):
        dev = ctx.device
        self.ops, self.ctx, self.pos, self.B, self.seed, self.enabled = ops, ctx, pos, B, seed, enabled
        # presort_state: a BprmfDeviceState -- the batch is also ORDERED (prep + radix sort, which read only the triplets) ahead
        # of its step, into one workspace per buffer
        self.state = presort_state if enabled else None
        self.ws = [presort_state.sort_workspace(B) for _ in range(2)] if self.state is not None else [None, None]
        self.bufs = [tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3)) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [None, None]
        self.cur, self.ctr = 0, 0
        if enabled:
            self._issue(0)

    def _draw(self, b):
        self.ops.bpr_sample(self.ctx, self.pos, self.B, seed=self.seed, first_sample=self.ctr, out=self.bufs[b])
        self.ctr += self.B
        if self.state is not None:
            self.state.presort(*self.bufs[b], self.ws[b])

    def _issue(self, b):
        self.side.wait_stream(torch.cuda.current_stream())           # (first use / anything the caller queued before)
        with torch.cuda.stream(self.side):
            if self.free[b] is not None:
                self.side.wait_event(self.free[b])                    # the training step that read this buffer is done
            self._draw(b)
            self.ready[b].record(self.side)

    def next(self):
        """Triplets of this step (valid on the current stream)."""
        b = self.cur
        if not self.enabled:
            self._draw(b)
            return self.bufs[b], b
        torch.cuda.current_stream().wait_event(self.ready[b])
        self.cur ^= 1
        self._issue(self.cur)                                         # next batch: overlaps this step's kernels
        return self.bufs[b], b

    def release(self, b):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.free[b] = ev


class PrefetchPointwise:
    """The same pipeline for the point-wise models (NeuMF / GMF: neural_matrix_factorization.py:87-95 `for batch in sampler.step(...):
    train_step(batch)`): (u, i, label) of step t+1 drawn on a side stream under step t, and -- presort_state: an NmfDeviceState --
    the batch's (embedding row, sample) keys ordered there too (el_nmf_presort: the sort reads u and i only)."""

    def __init__(self, ctx, pos, B, seed, enabled=True, presort_state=None):
        dev = ctx.device
        self.ctx, self.pos, self.B, self.seed, self.enabled = ctx, pos, B, seed, enabled
        self.state = presort_state if enabled else None
        self.bufs = [(torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                      torch.empty(B, dtype=torch.float32, device=dev)) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [None, None]
        self.cur, self.ctr = 0, 0
        if enabled:
            self._issue(0)

    def _draw(self, b):
        ops.pointwise_sample(self.ctx, self.pos, self.B, seed=self.seed, first_sample=self.ctr, out=self.bufs[b])
        self.ctr += self.B
        if self.state is not None:
            self.state.presort(self.bufs[b][0], self.bufs[b][1])

    def _issue(self, b):
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            if self.free[b] is not None:
                self.side.wait_event(self.free[b])                    # the training step that read this buffer is done
            self._draw(b)
            self.ready[b].record(self.side)

    def next(self):
        """(u, i, label) of this step (valid on the current stream) and the buffer index to release() after the step."""
        b = self.cur
        if not self.enabled:
            self._draw(b)
            return self.bufs[b], b
        torch.cuda.current_stream().wait_event(self.ready[b])
        self.cur ^= 1
        self._issue(self.cur)                                         # next batch: overlaps this step's kernels
        return self.bufs[b], b

    def release(self, b):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.free[b] = ev


def cover_triplets(indptr, indices, cover_users, cover_items, U, I, B):
    """The batches of cover_batches(), one (u, i, j) int32 triple per step: users in order with one of their positives each,
    negatives in item order."""
    dev = indptr.device
    n_cover = -(-max(cover_users, cover_items) // B)
    ar = torch.arange(B, dtype=torch.int64, device=dev)
    deg = indptr[1:] - indptr[:-1]
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    for c in range(n_cover):
        uu = (ar + c * B) % U
        off = (torch.rand(B, device=dev, generator=g) * deg[uu].to(torch.float32)).to(torch.int64)
        off = torch.minimum(off, deg[uu] - 1).clamp_(min=0)
        ii = indices[(indptr[uu] + off).clamp_(max=indices.numel() - 1)]
        jj = ((ar + c * B) % I).to(torch.int32)
        yield uu.to(torch.int32), ii.to(torch.int32), jj


def cover_batches(st, indptr, indices, cover_users, cover_items, U, I, B, lr, l_w, l_b, algo="auto"):
    """Untimed train steps that give the first `cover_users` user rows and the first `cover_items` item rows a gradient once (users in
    order with one of their positives each, negatives in item order): afterwards no row of the covered tables sits at the m = v = 0
    fixed point of the gradient-free Adam step, which the replay kernels of the deferred decay skip."""
    for uu, ii, jj in cover_triplets(indptr, indices, cover_users, cover_items, U, I, B):
        st.train_step(uu, ii, jj, lr, l_w, l_b, algo=algo)
    st.sync()
    st.pop_loss()
