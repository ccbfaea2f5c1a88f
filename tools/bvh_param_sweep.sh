#!/bin/bash
# host-builder parameter sweep: frame time of the default kernel over trees built with different SAH constants
for B in "" "sah_node_cost=0.5" "sah_node_cost=2" "sah_tri_cost=1.5" "sah_tri_cost=0.7" "split_alpha=1e-4" "split_alpha=1e-6" "n_bins=64" "sah_node_cost=0.5,split_alpha=1e-6"; do
  echo "== $B"
  timeout -k 10 120 python tools/sweep_gpu.py --bvh "$B" --spp 8 --frames 5 --rounds 3 --variants persist:36:0:6:16:2:8 2>&1 | grep -E "^bvh|persist" || exit 1
done
