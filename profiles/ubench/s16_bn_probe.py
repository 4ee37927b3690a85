"""Standalone rates of the 16-bit BatchNorm kernels on the model's shapes: python profiles/ubench/s16_bn_probe.py
colstats (column reduce, one tensor read), bn_apply (read + write), bn_bwd (reduce: 2 reads; apply: 2 reads + 1 write)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from glfusion_amd._lib import check, lib  # noqa: E402

DEV = "cuda"
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    st = torch.cuda.current_stream().cuda_stream
    # flush buffer: keep the tensors from sitting in the Infinity Cache between repetitions
    big = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    print(f"{'rows':>7s} {'c':>5s} | colstats us  TB/s | bn_apply us  TB/s | bn_bwd us  TB/s (5 tensor passes; incl. a 2c-double fill launch)")
    for rows, c in ((193600, 64), (193600, 256), (50176, 128), (50176, 256), (50176, 512), (50176, 1024), (50176, 2048)):
        x = torch.randn(rows, c, device=DEV).to(torch.bfloat16)
        dy = torch.randn(rows, c, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        sums = torch.zeros(2 * c, dtype=torch.float64, device=DEV)
        f = lambda: torch.empty(c, device=DEV)
        gamma, beta, mean, invstd, dg, db = torch.ones(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.ones(c, device=DEV), f(), f()
        mb = rows * c * 2 / 1e6

        def colstats():
            sums.zero_()
            check(lib.glf_s16_colstats(p(x), c, rows, c, p(sums), st), "colstats")

        def apply_():
            check(lib.glf_s16_bn_apply(p(x), c, None, 0, p(y), c, None, rows, c, 1e-5, 0.1, p(gamma), p(beta), p(mean), p(invstd), None, None, None, 1, None, st), "apply")

        def bwd():
            sums.zero_()
            check(lib.glf_s16_bn_bwd(p(dy), c, None, 0, p(x), c, p(mean), p(invstd), p(gamma), p(beta), p(dx), c, None, 0, p(dg), p(db), rows, c, 1, 1, p(sums), None, st), "bwd")

        t1, t2, t3 = timeit(colstats), timeit(apply_), timeit(bwd)
        print(f"{rows:7d} {c:5d} | {t1:8.1f} {mb / t1:6.2f} | {t2:8.1f} {2 * mb / t2:6.2f} | {t3:8.1f} {5 * mb / t3:6.2f}")
    del big


if __name__ == "__main__":
    main()
