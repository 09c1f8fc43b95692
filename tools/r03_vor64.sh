# 64-byte wall records (n, m precomputed) for the cluster-tiled Voronoi walk: tests, then A/B at 1e8 packets on the real tessellation
python -m pytest tests/test_gpu_voronoi.py -x -q 2>&1 | tail -3
for v in v512 v64 v64u2; do HYP_LIB=build/variants/$v.so python tools/voronoi_big_bench.py 1e8 one 2>&1 | tail -1 | cut -c1-170; done
HYP_LIB=build/variants/v64.so python tools/voronoi_big_bench.py 1e8 one vt_cells=100 2>&1 | tail -1 | cut -c1-170
python tools/voronoi_big_bench.py 1e8 2>&1 | tail -1 | cut -c1-170
