#!/bin/bash
# Per-round GPU run: the whole GPU test suite (as the driver runs it), parity metrics kept
O=gpurun_out/r${ROUND:-5}_tests; mkdir -p $O; rm -f gpurun_out/parity_metrics.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/tests.log
