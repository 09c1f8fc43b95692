# 512-thread (two workgroups per CU, 78 KB) against 1024-thread (one per CU, 156 KB) walk workgroups on the octree, AMR and
# Voronoi tiled schedules, same box, 1e8 packets (build/variants: GEOM=1|3|2 tools/variants.py build x512:"" x1024:"-DHYP_xTILE_WG=1024")
for o in "" "tile_task=8192"; do HYP_LIB=build/variants/o512.so python tools/octree_lucy.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
for o in "ot_lds_kb=156" "ot_lds_kb=156 tile_task=8192"; do HYP_LIB=build/variants/o1024.so python tools/octree_lucy.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
for o in "" "tile_task=8192"; do HYP_LIB=build/variants/a512.so python tools/amr_lucy.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
for o in "at_lds_kb=156" "at_lds_kb=156 tile_task=8192"; do HYP_LIB=build/variants/a1024.so python tools/amr_lucy.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
for o in "one" "tile_task=8192 one"; do HYP_LIB=build/variants/v512.so python tools/voronoi_big_bench.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
for o in "vt_lds_kb=156 one" "vt_lds_kb=156 tile_task=8192 one"; do HYP_LIB=build/variants/v1024.so python tools/voronoi_big_bench.py 1e8 $o 2>&1 | tail -1 | cut -c1-150; done
