#!/usr/bin/env python3
"""Bare v_mfma_f32_16x16x32_bf16 issue rate on this chip (operands in registers): the practical
ceiling the GEMM K loops are measured against."""
import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_peak.so"))
L.mfma_run.restype = ctypes.c_float
L.mfma_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256 * 8 * 256, device="cuda")
clk = torch.zeros(2, device="cuda", dtype=torch.int64)
for wps, nacc in ((1, 16), (2, 16), (4, 16), (2, 4), (1, 32), (2, 32)):
    blocks = 256 * wps
    iters = 20000
    ms = L.mfma_run(blocks, iters, nacc, out.data_ptr(), clk.data_ptr())
    c = clk.cpu().tolist()
    ghz = c[0] / max(c[1], 1) * 0.1
    if nacc == 32:      # v_mfma_f32_32x32x16_bf16, 4 independent accumulator tiles
        flops = blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16
        print("32x32x16: waves/SIMD %d: %.1f ms -> %.0f TFLOP/s, %.1f cycles per MFMA per SIMD" % (wps, ms, flops / ms / 1e9, ms * 1e-3 * (clk.cpu().tolist()[0] / max(clk.cpu().tolist()[1], 1) * 0.1) * 1e9 / (iters * 4 * wps)))
        continue
    flops = blocks * 4 * iters * nacc * 2.0 * 16 * 16 * 32
    print("waves/SIMD %d, %2d independent accumulators: %.1f ms -> %.0f TFLOP/s; shader clock while running %.2f GHz (s_memtime / s_memrealtime) -> "
          "%.1f cycles per MFMA per SIMD, dense peak at that clock %.0f TFLOP/s" % (
        wps, nacc, ms, flops / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / (iters * nacc * wps), 2500.0 * ghz / 2.4))
