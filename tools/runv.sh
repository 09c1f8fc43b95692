L=hyperion_amd/csrc/libhyperion_amd.so
python tools/variants.py one $L 1e8 lucy_mode=1 tile_prep_blocks=1
python tools/variants.py one $L 1e8 lucy_mode=1 tile_prep_blocks=1 tile_pools=4 tile_slots=8388608
python tools/variants.py one $L 1e8 lucy_mode=1 tile_prep_blocks=1 tile_pools=2 tile_slots=4194304
python tools/variants.py one $L 1e8 lucy_mode=1 tile_prep_blocks=1 tile_slots=9437184
python tools/variants.py one $L 1e8 lucy_mode=1 tile_prep_blocks=1 tile_slots=4718592
python tools/variants.py one $L 2e7 lucy_mode=1 tile_prep_blocks=1
python tools/variants.py one $L 2e7 lucy_mode=1 tile_prep_blocks=8
