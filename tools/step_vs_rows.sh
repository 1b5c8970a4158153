#!/bin/bash
# Step time and per-kernel HIP-event times of the product step against the rows per step (5,000 cells): what part of a
# step is a fixed floor (six short kernels, each at least one wave-lifetime long) and what part grows with the rows.
# On the GPU box: bash tools/step_vs_rows.sh   (appends to gpurun_out/sizes.txt; tools/two_slot_perf.py ROWS gives the
# same step without the timing events, one slot and two slots in flight)
for n in 125000 250000 500000 1000000 2000000 4000000; do
  timeout -k 10 100 python tools/quick_perf.py $n 5000 2>&1 | grep -E "mode=0 shortcut=1" >> gpurun_out/sizes.txt || exit 1
done
