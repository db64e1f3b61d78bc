#!/usr/bin/env python3
"""Fills (L2 -> LDS DMA, or L2 -> registers) and ds_read_b128 streams of other waves on the same CU:
alone and together (tools/micro/lds_mix.hip)."""
import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_mix.so"))
L.mix_run.restype = ctypes.c_float
L.mix_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(4096, device="cuda")
nbytes = 2 << 20
src = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
for occ in (1, 2):
    lds = (160 * 1024 // occ) // 1024 * 1024
    blocks = 256 * occ
    fill_iters = 600
    for to_regs in (0, 1):
        tf = L.mix_run(to_regs, src.data_ptr(), nbytes, fill_iters, 0, 1, blocks, lds, sink.data_ptr())
        fill_rate = blocks * fill_iters * 32768.0 / (tf * 1e-3 * 2.4e9 * 256)
        # reads sized to take about as long as the fills: 2 x the fill bytes (the GEMM reads 64 KB per 32 KB filled)
        for ratio in (2, 4):
            read_iters = fill_iters * ratio * 32768 // (4 * 8 * 1024)
            tr = L.mix_run(to_regs, src.data_ptr(), nbytes, 0, read_iters, 2, blocks, lds, sink.data_ptr())
            tb = L.mix_run(to_regs, src.data_ptr(), nbytes, fill_iters, read_iters, 3, blocks, lds, sink.data_ptr())
            read_rate = blocks * read_iters * 4 * 8 * 1024.0 / (tr * 1e-3 * 2.4e9 * 256)
            print("%d wg/CU, fills to %s: fills alone %.3f ms (%.1f B/clk/CU), reads(x%d bytes) alone %.3f ms (%.0f B/clk/CU), together %.3f ms (max %.3f, sum %.3f)" % (
                occ, "regs" if to_regs else "LDS ", tf, fill_rate, ratio, tr, read_rate, tb, max(tf, tr), tf + tr), flush=True)
