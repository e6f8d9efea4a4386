#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/hodlr; export TMPDIR=/tmp
timeout -s KILL 300 python scripts/dev/hodlr_singles_ab.py > gpurun_out/hodlr/singles_ab.md 2>&1; cat gpurun_out/hodlr/singles_ab.md
LINES_OUT=12 SIZES=262144 bash scripts/gpu_hodlr_quick.sh
