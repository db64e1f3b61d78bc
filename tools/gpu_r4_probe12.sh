#!/bin/bash
mkdir -p gpurun_out/p12
MMTG_EXTRA_DEFS=-DMMTG_P8_PHASE_TRACE python -m mmtg_amd.build --force --jobs 16 2>&1 | tail -1
( echo "== priority raise around the MFMA clusters (default)"; timeout 300 python tools/p8_phase_trace.py
  echo "== MMTG_P8_NOPRIO=1"; MMTG_P8_NOPRIO=1 timeout 300 python tools/p8_phase_trace.py ) 2>&1 | grep -v amdgpu > gpurun_out/p12/p8_phase_trace.txt; cat gpurun_out/p12/p8_phase_trace.txt
python -m mmtg_amd.build --force --jobs 16 2>&1 | tail -1
( echo "== TNSET eight-phase K-strided slabs, default"; MMTG_GEMM_P8T=1 SLAB=2 TNSET=1 python tools/bench_gemm.py; echo "== NOPRIO"; MMTG_P8_NOPRIO=1 MMTG_GEMM_P8T=1 SLAB=2 TNSET=1 python tools/bench_gemm.py
  echo "== NTSET default"; NTSET=1 python tools/bench_gemm.py; echo "== NTSET NOPRIO"; MMTG_P8_NOPRIO=1 NTSET=1 python tools/bench_gemm.py ) 2>&1 | grep -v amdgpu > gpurun_out/p12/noprio_ab.txt; cat gpurun_out/p12/noprio_ab.txt
