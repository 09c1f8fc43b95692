for f in build/variants/o_*.so; do for o in "tile_slots=12582912" "tile_slots=25165824 tile_task=2048"; do HYP_LIB=$f python tools/octree_lucy.py 1e8 $o 2>&1 | grep -v amdgpu; done; done
