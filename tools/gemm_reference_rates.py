#!/usr/bin/env python3
"""What the vendor GEMM reaches on the Res5 head's GEMM shapes (measurement only: torch.matmul = hipBLASLt / rocBLAS; never
on the product path).  Random fp16 data, fp32 accumulate, no bias / residual / epilogue, 20 launches after warm-up.
GPU box only.  usage: python tools/gemm_reference_rates.py"""
import torch

SHAPES = {   # name: (M, N, K)
    "head conv1 (2048->512)": (1881600, 512, 2048),
    "head conv1 block0 (1024->512)": (1881600, 512, 1024),
    "head conv3 (512->2048)": (1881600, 2048, 512),
    "head conv3+shortcut (1536->2048)": (1881600, 2048, 1536),
    "head conv2 as im2col GEMM (4608->512)": (1881600 // 4, 512, 4608),
    "res4 conv1 (1024->256)": (134400, 256, 1024),
    "res4 conv3 (256->1024)": (134400, 1024, 256),
    "square 8192": (8192, 8192, 8192),
}


def main():
    dev = torch.device("cuda:0")
    for name, (M, N, K) in SHAPES.items():
        a = torch.randn((M, K), device=dev, dtype=torch.float16)
        b = torch.randn((N, K), device=dev, dtype=torch.float16)
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:42s} M={M:8d} N={N:5d} K={K:5d}  {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)
        del a, b, c


if __name__ == "__main__":
    main()
