python tools/variants.py one build/variants/pstats.so 1e8 lucy_mode=1 tile_pools=1 tile_slots=8388608 2>&1 | grep "lib\|prepare stats" | tail -2
