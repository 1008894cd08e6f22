#!/bin/bash
O=gpurun_out/r4_2; mkdir -p $O
python tools/dbg/rds_sym_stats.py 24 32 > $O/sym_sparse.log 2>&1
FMD_DEBUG_PLL_DENSE=1 python tools/dbg/rds_sym_stats.py 24 32 > $O/sym_dense.log 2>&1
