# record prefetch of tile_walk (HYP_TILE_PREFETCH): parity of the default build, then look-aheads 0 / 128 / 256 / 512 at 1e8 packets
python tools/tiled_check.py 2>&1 | grep -v "^$" | head -12
for v in pf0 pf128 pf256 pf512; do for o in "" "tile_pools=1"; do python tools/variants.py one build/variants/$v.so 1e8 $o 2>&1 | tail -1 | cut -c1-170; done; done
