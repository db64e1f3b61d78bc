#!/usr/bin/env python3
"""L2 -> LDS fill rate per CU by access shape / workgroups per CU / tiles in flight (tools/micro/lds_fill.hip)."""
import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_fill.so"))
L.fill_run.restype = ctypes.c_float
L.fill_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(4096, device="cuda", dtype=torch.int32)
iters = 400
names = {0: "8 rows x 128 B (K-contiguous rows)", 1: "4 rows x 256 B (K-strided rows)", 2: "1 KB contiguous (pre-tiled)"}
for mb in (2, 24):          # footprint: inside every XCD's L2 / inside the Infinity Cache only
    for ld in (1536, 6144):
        rows = mb * (1 << 20) // ld
        nbytes = rows * ld
        src = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
        for shape in (0, 1, 2):
            if shape == 2 and ld != 1536:
                continue
            for depth in (1, 2):
                line = []
                for occ in (1, 2, 3, 4):
                    if depth * 32768 * occ > 160 * 1024:
                        continue
                    lds = max(depth * 32768, (160 * 1024 // occ) & ~1023) if occ > 1 else 160 * 1024
                    lds = min(lds, 160 * 1024)
                    # LDS per workgroup chosen so that exactly `occ` workgroups fit a CU
                    lds = (160 * 1024 // occ) // 1024 * 1024
                    if (160 * 1024) // lds != occ:
                        lds -= 1024
                    blocks = 256 * occ
                    ms = L.fill_run(shape, depth, src.data_ptr(), nbytes, ld, rows, iters, blocks, lds, sink.data_ptr())
                    tot = blocks * iters * 32768.0
                    line.append("%d/CU: %5.1f B/clk/CU %5.1f TB/s" % (occ, tot / (ms * 1e-3 * 2.4e9 * 256), tot / ms / 1e9))
                print("%3d MB ld=%5d %-36s depth %d | %s" % (mb, ld, names[shape], depth, " | ".join(line)), flush=True)
