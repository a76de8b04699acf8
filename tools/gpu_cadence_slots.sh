#!/bin/bash
for s in 2 3 4 5 6 8; do echo "slots $s"; CADENCE_CONFIGS=headline GLRTX_PIPE_SLOTS=$s timeout -k 10 120 python tools/gpu_cadence.py 20 100 300 2>/dev/null; done
