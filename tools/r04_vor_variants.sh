#!/bin/bash
# time every tuning variant of the Voronoi walk in build/variants (tools/variants.py build ...; GEOM=2 ND=2) on configs[4]
#   usage: bash tools/r04_vor_variants.sh [packets] [name:"opt=value ..."] ...   (default: every .so, no options)
N=${1:-1e8}; shift
run() { echo "== $1 $2"; HYP_LIB=build/variants/$1.so python tools/voronoi_big_bench.py $N $2 2>&1 | grep -v amdgpu.ids | tail -${3:-1} | cut -c1-75,96-330; }
if [ $# -eq 0 ]; then for f in build/variants/*.so; do run $(basename $f .so) ""; done
else for spec in "$@"; do run "${spec%%:*}" "$( [[ $spec == *:* ]] && echo ${spec#*:})" ; done; fi
