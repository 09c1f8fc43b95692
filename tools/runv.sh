L=hyperion_amd/csrc/libhyperion_amd.so
python tools/variants.py one $L 1e8 lucy_mode=1 tile_slots=3145728 tile_drain=1000000
python tools/variants.py one $L 1e8 lucy_mode=1 tile_slots=4718592 tile_drain=1000000
python tools/variants.py one $L 1e8 lucy_mode=1 tile_slots=6291456 tile_pools=4 tile_drain=1000000
python tools/variants.py one $L 1e8 lucy_mode=1 tile_slots=6291456 tile_drain=500000
python tools/variants.py one $L 1e8 lucy_mode=1 tile_slots=6291456 tile_drain=1000000 tile_task=2048
