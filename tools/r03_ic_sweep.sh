# tile_interact chunking / occupancy variants + a timeline of the default build (configs[1], 1e8 packets)
for v in ic512 ic256 iw3; do python tools/variants.py one build/variants/$v.so 1e8 2>&1 | tail -1 | cut -c1-170; done
bash tools/r03_timeline.sh tl_car python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -30
