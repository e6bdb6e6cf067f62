# timing-only ablation builds of conv_ws_kernel<16> (VK_WS_DBG: 1 no epilogue (the MFMAs become dead code too), 2 no MFMA,
# 4 no pixel DMA, 8 no residual DMA; WRONG results; 16 = the first version's stage waits, correct) on the Res5 conv3 shape
# Needs the tools build of the library: make -C vltk_amd/csrc clean && make -C vltk_amd/csrc -j8 ABLATION=1 (the shipped build ignores VK_WS_DBG).
# (512 -> 2048, M = 200 704), with one (VK_WS_WAVES=4) or two (8) waves per SIMD.  GPU box only.
for w in ${WAVES:-4 8}; do for d in 0 1 2 3 4 8 12 15; do echo "waves=$w dbg=$d"; VK_WS_WAVES=$w VK_WS_DBG=$d timeout -k 10 100 python tools/conv_bench.py head_conv3; done; done
