"""Experiment: does the relative placement of theta / g / m / v in HBM change the dense-Adam pass?  The four [U, F] arrays are
carved from one buffer with a configurable gap between them; el_bprmf_apply (TF-dense Adam) is timed with the library's
own hipEvent timers."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops                      # noqa: E402
from elliot_amd._lib import BprmfState          # noqa: E402


def main():
    ctx = ops.get_context(0)
    dev = ctx.device
    U, I, F = 1_000_000, 1024, 128
    n = U * F
    gaps = [int(x) for x in os.environ.get('GAPS', '').split(',') if x] or [0, 3 << 20]
    for gap in gaps:
        tot = 4 * (n * 4 + gap) + (1 << 22)
        buf = torch.zeros(tot, dtype=torch.uint8, device=dev)
        base = (buf.data_ptr() + 255) & ~255
        ptr = [base + t * (n * 4 + gap) for t in range(4)]
        small = [torch.zeros((I, F), device=dev) for _ in range(4)] + [torch.zeros(I, device=dev) for _ in range(4)]
        st = BprmfState(Gu=ptr[0], gGu=ptr[1], mGu=ptr[2], vGu=ptr[3], Gi=small[0].data_ptr(), gGi=small[1].data_ptr(),
                        mGi=small[2].data_ptr(), vGi=small[3].data_ptr(), Bi=small[4].data_ptr(), gBi=small[5].data_ptr(),
                        mBi=small[6].data_ptr(), vBi=small[7].data_ptr(), tGu=None, tGi=None, tBi=None, U=U, I=I, F=F)
        for it in range(3):
            ops.check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(st), 0.001, 0, it + 1, 0.001), "apply")
        torch.cuda.synchronize()
        ctx.timing(True)
        for it in range(10):
            ops.check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(st), 0.001, 0, it + 4, 0.001), "apply")
        torch.cuda.synchronize()
        rep = ctx.timing_report()
        ctx.timing(False)
        ms = rep["k_adam_dense_Gu"][1] / rep["k_adam_dense_Gu"][0]
        print(f"gap {gap:>10d} B  base%2MB={base % (2 << 20):>8d}  k_adam_dense_Gu {ms:.3f} ms  ({24.0 * n / ms / 1e6:.0f} GB/s algorithmic)")
        del buf


if __name__ == "__main__":
    main()
