#!/bin/bash
timeout 120 python tools/micro/mfma_peak.py 2>&1 | grep -v amdgpu
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head
