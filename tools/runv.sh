for v in t256 t512 t1024 t512f t512g t256g; do
  python tools/variants.py one build/variants/$v.so 1e8 lucy_mode=1 tile_slots=8388608
done
python tools/variants.py one build/variants/t512.so 1e8 lucy_mode=1 tile_slots=16777216
python tools/variants.py one build/variants/t512g.so 1e8 lucy_mode=1 tile_slots=16777216
python tools/variants.py one build/variants/t512.so 1e8 lucy_mode=1 tile_slots=8388608 tile_task=2048
python tools/variants.py one build/variants/t1024.so 1e8 lucy_mode=1 tile_slots=8388608 tile_task=8192
