L=hyperion_amd/csrc/libhyperion_amd.so
python tools/variants.py one $L 1e8 lucy_mode=1
for v in b20 b16x32 b16w1024 g32 g24; do python tools/variants.py one build/variants/$v.so 1e8 lucy_mode=1; python tools/variants.py one build/variants/$v.so 1e8 lucy_mode=1 tile_task=8192; done
