L=hyperion_amd/csrc/libhyperion_amd.so
python tools/variants.py one $L 1e8 lucy_mode=1 tile_pools=1 tile_slots=8388608
for pk in 0 8 16 24 32 48; do python tools/variants.py one $L 1e8 lucy_mode=1 tile_park=$pk; done
python tools/variants.py one $L 1e8 lucy_mode=1 tile_task=8192
python tools/variants.py one $L 2e7 lucy_mode=1
python tools/variants.py one $L 4e6 lucy_mode=1
