#!/bin/bash
# round 6: every conv / forward / benched-config / SSD parity test with the tile-order panel width FORCED (RON_PANEL_COLS): a tile order
# that skipped or repeated a tile could not pass them.  The plan assertion of test_panel_tile_orders_cover_every_tile is deselected (it
# asserts the default choice).
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_panel_parity
mkdir -p $O
for P in 3 5 2; do
  RON_PANEL_COLS=$P timeout 1200 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_forward.py tests/test_gpu_benched_config.py tests/test_gpu_ssd.py tests/test_gpu_pipeline.py -m gpu -q --deselect tests/test_gpu_conv.py::test_panel_tile_orders_cover_every_tile > $O/pytest_P$P.txt 2>&1
  echo "RON_PANEL_COLS=$P: $(grep -E 'passed|failed' $O/pytest_P$P.txt | tail -1)"
done
