L=hyperion_amd/csrc/libhyperion_amd.so
python tools/variants.py one $L 1e8 lucy_mode=1 tile_pools=1 tile_slots=8388608
python tools/variants.py one $L 1e8 lucy_mode=1
