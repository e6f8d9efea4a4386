#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 300 python scripts/dev/gemm_sp_ab.py > gpurun_out/gemm_sp_ab.log 2>&1; echo "ab rc=$?"; tail -20 gpurun_out/gemm_sp_ab.log
