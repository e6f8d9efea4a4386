#!/bin/bash
bash scripts/gpu_hodlr_tests.sh
LINES_OUT=14 SIZES="262144 50000 1048576" bash scripts/gpu_hodlr_quick.sh
