#!/bin/bash
# balanced task sizes (tile_balance, tile_task_min) against fixed ones on the tiled Lucy schedules, 1e8 packets
for o in "tile_balance=0" "tile_balance=1" "tile_balance=1 tile_task_min=3072" "tile_balance=1 tile_task_min=4096" "tile_balance=1 tile_task_min=1024"; do
  echo "== $o"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $(for x in $o; do echo --option $x; done) 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  python tools/voronoi_big_bench.py 1e8 $o 2>&1 | tail -1 | grep -o "kernel [0-9.]* ms" | sed 's/^/vor /'
  python tools/r03_workload.py oct_lucy 1e8 $o 2>&1 | grep "^oct_lucy" | tail -1 | grep -o "kernel [0-9.]* ms" | sed 's/^/oct /'
  python tools/r03_workload.py amr 1e8 $o 2>&1 | grep "^amr" | tail -1 | grep -o "kernel [0-9.]* ms" | sed 's/^/amr /'
done
