# register budget of the imaging kernels (HYP_FINAL_WAVES = waves per SIMD the budget is set for): configs[3] imaging at 1e8 packets and
# the general kernel / raytracing iterations at 1e7 on octree-only variants (GEOM=1 python tools/variants.py build base: fw1:"-DHYP_FINAL_WAVES=1" ...)
for v in ${VARIANTS:-base fw1 fw3}; do
  echo "== $v"
  HYP_LIB=build/variants/$v.so python tools/r03_workload.py oct_img ${PACKETS:-1e8} 2>&1 | grep "^oct_img" | cut -c1-200
  HYP_LIB=build/variants/$v.so python tools/r03_other.py 1e7 2>&1 | grep "^| " | cut -c1-200
done
