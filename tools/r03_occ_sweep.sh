# walk-kernel occupancy variants of the Cartesian tiled schedule (tools/variants.py build ...), configs[1] at 1e8 packets
for v in base occ8_16 occ8_32nd occ6_16; do for o in "" "tile_task=4096"; do python tools/variants.py one build/variants/$v.so 1e8 $o 2>&1 | tail -1 | cut -c1-170; done; done
