#!/usr/bin/env python3
"""Bare v_mfma_f32_16x16x32_bf16 issue rate on this chip (operands in registers): the practical
ceiling the GEMM K loops are measured against."""
import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_peak.so"))
L.mfma_run.restype = ctypes.c_float
L.mfma_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 8 * 256, device="cuda")
for wps, nacc in ((1, 16), (2, 16), (4, 16), (2, 4)):
    blocks = 256 * wps
    iters = 20000
    ms = L.mfma_run(blocks, iters, nacc, out.data_ptr())
    flops = blocks * 4 * iters * nacc * 2.0 * 16 * 16 * 32
    print("waves/SIMD %d, %2d independent accumulators: %.1f ms -> %.0f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)" % (
        wps, nacc, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * nacc * wps)))
