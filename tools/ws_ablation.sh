# timing-only ablation builds of conv_ws_kernel<16> (VK_WS_DBG: 1 no epilogue (the MFMAs become dead code too), 2 no MFMA,
# 4 no pixel DMA, 8 no residual DMA; WRONG results) on the Res5 conv3 shape (512 -> 2048, M = 200 704).  GPU box only.
for d in 0 1 2 3 4 8 12 15; do echo "dbg=$d"; VK_WS_DBG=$d timeout -k 10 100 python tools/conv_bench.py head_conv3; done
